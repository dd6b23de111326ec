cd /root/repo
for G in 4 5 6 8; do
SGMCMC_EXACT_LANES=3 SGMCMC_EXACT_GROUP=$G python tools/exact_pass_probe.py --passes 4 2>&1 | tail -1
done
SGMCMC_EXACT_LANES=2 SGMCMC_EXACT_GROUP=6 python tools/exact_pass_probe.py --passes 4 2>&1 | tail -1
SGMCMC_EXACT_LANES=4 SGMCMC_EXACT_GROUP=6 python tools/exact_pass_probe.py --passes 4 2>&1 | tail -1
