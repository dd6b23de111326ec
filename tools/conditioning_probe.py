"""How well conditioned is googleresnet's exact full-data gradient at batch 128, and where?  At several points of a
VerletSGLDReject run (initialisation, after K leapfrog steps) the product's default exact pass (3 lanes x G minibatches
per launch, persistent convolutions) is compared with a FLOAT64 autograd evaluation of the same quantity at the same
parameters (the same modules cast to double: every kernel of this package declines float64, so that pass runs on the
library's double kernels) and with float32 on the library's kernels (evaluation._plain_torch_layers) as the yard-stick of what float32
itself costs at that point:  max |g - g64| / max |g64| per parameter tensor, worst tensor reported.
    python tools/conditioning_probe.py [--steps 0,50,200,400]"""
import argparse
import copy
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import runner_cases as RC
from bnn_priors_amd import inference_reject, models
from bnn_priors_amd.storage import MemoryMetrics


def f64_gradient(model, batches, N, dtype=torch.float64):
    "autograd gradient of  sum_batches[-sum_i log p_i / N] - log_prior / N  (inference_reject.py:18-33) in ``dtype``"
    ref = copy.deepcopy(model).to(dtype)
    ref.train()
    ref.zero_grad()
    (ref.log_prior() / -N).backward()
    loss = 0.0
    for x, y in batches:
        this = ref.log_likelihood(x.to(dtype), y, -x.size(0) / N)
        this.backward()
        loss += float(this)
    return [p.grad.double() for p in ref.parameters()], loss


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", default="0,50,200,400")
    ap.add_argument("--n", type=int, default=1024)
    a = ap.parse_args()
    dev = "cuda:0"
    g = torch.Generator().manual_seed(21)
    x = torch.randn((a.n, 3, 32, 32), generator=g).to(dev)
    y = torch.randint(0, 10, (a.n,), generator=g).to(dev)
    train = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(x, y), batch_size=128, shuffle=False)
    test = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(x[:128], y[:128]), batch_size=128)
    model = RC.make_net(models, x[:2].cpu(), torch.tensor([0, 9]), device=dev, cfg=dict(model="googleresnet"))
    torch.manual_seed(RC.SEED)
    runner = inference_reject.VerletSGLDRunnerReject(
        model=model, dataloader=train, dataloader_test=test, learning_rate=0.01, temperature=1.0, momentum=0.994,
        reject_samples=True, metrics_saver=MemoryMetrics(), model_saver=None, seed=RC.SEED, chain_id=0,
        cycle_seed=RC.CYCLE_SEED, use_graph=True, epochs_per_cycle=50, warmup_epochs=45, sample_epochs=5, skip=1,
        metrics_skip=10, cycles=60, precond_update=1, sampling_decay="cosine")
    step = runner.begin()
    pot = runner._potential()
    source = runner._batches()
    batches = list(train)
    done = 0
    for target in [int(s) for s in a.steps.split(",")]:
        while done < target:
            for xb, yb in runner._hot_batches():
                if done >= target:
                    break
                step += 1
                runner.leapfrog(step, xb, yb, last_of_epoch=False)
                done += 1
        runner._drain_rows()
        torch.cuda.synchronize()
        buffers = {k: v.clone() for k, v in model.state_dict().items() if "running" in k or "tracked" in k}
        g64, l64 = f64_gradient(model, batches, pot.N)
        from bnn_priors_amd import evaluation
        with evaluation._plain_torch_layers():                  # float32 on the LIBRARY's kernels, one minibatch at a time
            g32, l32 = f64_gradient(model, batches, pot.N, torch.float32)
        loss, _, _ = pot.exact(source)
        torch.cuda.synchronize()
        gp = [p.grad.double().clone() for p in pot.opt.engine.params]
        with torch.no_grad():
            for k, v in model.state_dict().items():
                if k in buffers:
                    v.copy_(buffers[k])

        def worst(ga):
            errs = [((u - v).abs().max() / v.abs().max().clamp_min(1e-300)).item() for u, v in zip(ga, g64)]
            i = max(range(len(errs)), key=errs.__getitem__)
            tot = (torch.cat([(u - v).flatten() for u, v in zip(ga, g64)]).abs().max()
                   / torch.cat([v.flatten() for v in g64]).abs().max()).item()
            return errs[i], i, tot
        wp, ip, tp = worst(gp)
        w3, i3, t3 = worst(g32)
        print(f"after {done:4d} steps: loss f64 {l64:.6f} product {loss.item():.6f} | product exact pass vs f64: worst tensor "
              f"{wp:.2e} (#{ip}), whole gradient {tp:.2e} | float32 library kernels vs f64: "
              f"worst {w3:.2e} (#{i3}), whole {t3:.2e}", flush=True)


if __name__ == "__main__":
    main()
