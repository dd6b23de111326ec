cd /root/repo
for P in 0 1; do for L in 1 3; do
SGMCMC_ALTERNATIVES=1 SGMCMC_CONV_PERSISTENT=$P SGMCMC_EXACT_LANES=$L SGMCMC_EXACT_GROUP=4 python tools/exact_pass_probe.py --passes 4 2>&1 | tail -1 | sed "s/^/persistent=$P: /"
done; done
SGMCMC_ALTERNATIVES=1 SGMCMC_CONV_PERSISTENT=1 SGMCMC_EXACT_LANES=3 SGMCMC_EXACT_GROUP=8 python tools/exact_pass_probe.py --passes 4 2>&1 | tail -1 | sed "s/^/persistent=1: /"
