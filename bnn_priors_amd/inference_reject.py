"""Exact-gradient Metropolis-Hastings drivers: ``VerletSGLDRunnerReject``,
``HMCRunnerReject``, ``SGLDRunnerReject``.  Drop-in for
``bnn_priors/inference_reject.py`` (reference :11-198): minibatch leapfrog steps
inside an epoch, a full-data gradient + ``final_step`` + M-H test at every stored
sample, one fixed minibatch order per cycle.

``leapfrog()`` is the hot-loop body (reference :86-113) and what ``bench.py`` times.
"""
import torch

from . import mcmc
from .inference import SGLDRunner, _is_hmc

__all__ = ("VerletSGLDRunnerReject", "HMCRunnerReject", "SGLDRunnerReject")


def _f(v):
    "python float of a number or 0-d tensor"
    return v.item() if isinstance(v, torch.Tensor) else float(v)


class VerletSGLDRunnerReject(SGLDRunner):
    def __init__(self, *a, cycle_seed=None, **kw):
        """``cycle_seed``: None reproduces the reference (a fresh non-deterministic
        shuffle seed per cycle, inference_reject.py:72); an int pins cycle c's seed to
        ``cycle_seed + c`` so that runs (and parity tests) are repeatable."""
        super().__init__(*a, **kw)
        self.cycle_seed = cycle_seed

    def _make_optimizer(self, params):
        return mcmc.VerletSGLD(params=params, lr=self.learning_rate, num_data=self.eff_num_data,
                               momentum=self.momentum, temperature=self.temperature,
                               **self._sampler_kwargs())

    def _exact_model_potential_and_grad(self, batches):
        """g <- grad[-log_prior/N] + sum_batches grad[-sum_i log p_i / N], accumulated over the
        whole loader (inference_reject.py:18-33); see potential.py."""
        fused = self._fused_dense()
        if fused is not None and getattr(batches, "fast", False) and batches.x is fused.X_source:
            return fused.exact()
        return self._potential().exact(batches)

    def leapfrog(self, step, x, y, last_of_epoch):
        """One minibatch leapfrog step: stochastic gradient, fused sampler transition,
        metrics every ``metrics_skip`` steps, LR schedule (inference_reject.py:86-113)."""
        store = (step % self.metrics_skip) == 0
        opt = self.optimizer
        def log_row(r, step=step, lr=opt.param_groups[0]["lr"], u0=self._initial_potential,
                    e0=self._total_energy):
            de = opt.delta_energy_from_total(r["energy"], u0, r["potential"])
            self._last_acc = r["acc"]
            self.store_metrics(i=step, loss=r["loss"], log_prior=r["log_prior"],
                               potential=r["potential"], acc=r["acc"], lr=lr,
                               corresponds_to_sample=False, delta_energy=de, total_energy=e0 + de)
        # quirk 5: the sample row logs the LAST minibatch's accuracy, so that one is always wanted
        handled, x, y = self._fast_plain_step(x, y, store, log_row, want_acc=last_of_epoch)
        if handled:
            if not last_of_epoch:
                self.scheduler.step()
                return None
            self._drain_rows()
            return self._last_acc
        loss, log_prior, potential, acc = self._model_potential_and_grad(x, y, store or last_of_epoch)
        opt.step(calc_metrics=store)
        if store:
            self._check_finite()
            delta_energy = self._delta_energy(potential)
            self.store_metrics(i=step, loss=loss.item(), log_prior=log_prior.item(),
                               potential=potential.item(), acc=acc.item(),
                               lr=opt.param_groups[0]["lr"],
                               corresponds_to_sample=False, delta_energy=_f(delta_energy),
                               total_energy=self._total_energy + _f(delta_energy))
        if not last_of_epoch:   # the last scheduler step of an epoch follows final_step
            self.scheduler.step()
        return acc

    def _delta_energy(self, potential):
        """energy difference since the last initial step, for the gradient the LAST transition
        used (true at both call sites: right after ``step`` / ``final_step``) -- the fused launch
        has already reduced it; samplers without that shortcut use ``delta_energy``"""
        fn = getattr(self.optimizer, "delta_energy_of_last_transition", self.optimizer.delta_energy)
        return fn(self._initial_potential, potential)

    def _trajectory_ends(self, last_of_epoch, epoch):
        """hook for runners that end trajectories inside an epoch (HMCRunnerReject(trajectory_length=L)): called once
        per leapfrog step BEFORE it runs; True = this step is followed by an M-H point of its own"""
        return False

    def _mh_point(self, step, acc, batches, save=None):
        """End of a trajectory: exact full-data gradient, ``final_step``, energy difference, Metropolis-Hastings
        test, metrics row; with ``save = (cycle, epoch)`` also evaluation + stored sample; then the momentum
        refresh (HMC) and the ``initial_step`` of the next trajectory on the same gradient with the next learning
        rate (inference_reject.py:115-157).  Consumes one step index (quirk 6).  Returns the step counter."""
        opt = self.optimizer
        step += 1                                                    # quirk 6
        loss, log_prior, potential = self._exact_model_potential_and_grad(batches)
        opt.final_step(calc_metrics=True)
        delta_energy = _f(self._delta_energy(potential))
        self._total_energy += delta_energy
        self._initial_potential = potential.item()                   # quirk 1
        rejected = False
        if self.reject_samples:
            rejected, _ = opt.maybe_reject(delta_energy)
        self._check_finite()
        self.store_metrics(i=step, loss=loss.item(), log_prior=log_prior.item(),
                           potential=potential.item(), acc=_f(acc),  # quirk 5
                           lr=opt.param_groups[0]["lr"],
                           corresponds_to_sample=save is not None, delta_energy=delta_energy,
                           total_energy=self._total_energy, rejected=rejected)
        if save is not None:
            state_dict = self.model.state_dict()
            self._evaluate_model(state_dict, step)
            self._save_sample(state_dict, save[0], save[1], step)
        self.scheduler.step()
        # first step of the next trajectory: same gradient, next learning rate (:152-157)
        if _is_hmc(opt):
            opt.sample_momentum()
        opt.initial_step(calc_metrics=False, save_state=self.reject_samples)
        return step

    def begin(self):
        """optimizer, scheduler, exact initial gradient, momentum draw and the first
        ``initial_step`` (inference_reject.py:36-66); returns the step counter (0)."""
        self.optimizer = opt = self._make_optimizer(self._params)
        self.scheduler = self._make_scheduler(opt)
        loss, log_prior, potential = self._exact_model_potential_and_grad(self._batches())
        opt.sample_momentum()
        opt.initial_step(calc_metrics=True, save_state=self.reject_samples)
        self._check_finite()
        self.store_metrics(i=0, loss=loss.item(), log_prior=log_prior.item(),
                           potential=potential.item(), acc=0.,                      # quirk 5
                           lr=opt.param_groups[0]["lr"], corresponds_to_sample=True,
                           delta_energy=0., total_energy=0., rejected=False)
        self._initial_potential = potential.item()
        self._total_energy = 0.
        return 0

    def run_iter(self):
        "inference_reject.py:35-179 as a generator (see SGLDRunner.run_iter): yields after every leapfrog step / epoch end"
        step = self.begin()
        opt, batches = self.optimizer, self._batches()

        def enter_epoch(temperature):
            for g in opt.param_groups:
                g['temperature'] = temperature

        assert self.dataloader.sampler.generator is None
        generator = self.dataloader.sampler.generator = torch.Generator()
        acc = torch.zeros(())
        try:
            for cycle in range(self.cycles):
                if self.cycle_seed is None:
                    generator.seed()
                else:
                    generator.manual_seed(self.cycle_seed + cycle)
                cycle_random_state = generator.get_state()
                for epoch in range(self.epochs_per_cycle):
                    if epoch < self.descent_epochs:
                        enter_epoch(0.)
                    else:
                        enter_epoch(self.temperature)
                    # same minibatch order in every epoch of the cycle (:84)
                    generator.set_state(cycle_random_state)
                    n_batches = len(batches)
                    lr_stepped = False
                    for i, (x, y) in enumerate(self._hot_batches()):
                        step += 1
                        last = i == n_batches - 1
                        # a step that ends a trajectory is treated like the last one of an epoch: the learning
                        # rate advances AFTER the final_step that closes it (:146), exactly once per minibatch
                        ends = self._trajectory_ends(last, epoch)
                        acc = self.leapfrog(step, x, y, last_of_epoch=last or ends)
                        if ends:
                            self._drain_rows()
                            step = self._mh_point(step, acc, batches)
                            lr_stepped = last
                        yield step

                    self._drain_rows()
                    if self._is_sampling_epoch(epoch):
                        step = self._mh_point(step, acc, batches, save=(cycle, epoch))
                    else:
                        self._evaluate_model(self.model.state_dict(), step)
                        if not lr_stepped:      # (an M-H point on the epoch's last step already advanced it)
                            self.scheduler.step()

                    if self.precond_update is not None and (epoch + 1) % self.precond_update == 0:
                        opt.update_preconditioner()
                    self._check_finite()
                    self.metrics_saver.flush(every_s=30)
                    yield step
        finally:
            self.dataloader.sampler.generator = None


class HMCRunnerReject(VerletSGLDRunnerReject):
    """inference_reject.py:182-189, plus two keyword-only EXTENSIONS for BASELINE.json configs[4] (the reference has
    neither: its trajectories are one epoch long and mcmc/hmc.py:39 asserts T == 1):

    ``trajectory_length=L``: a trajectory ends -- exact gradient, ``final_step``, M-H test, momentum refresh,
    ``initial_step`` -- after every L leapfrog steps as well as at the end of every sampling epoch (where the sample
    is stored, as in the reference).  Intra-epoch M-H points log a metrics row with ``acceptance/is_sample = 0``.
    The learning-rate schedule advances once per minibatch whatever L is (the step that ends a trajectory hands its
    scheduler step to the M-H point, as an epoch's last step does in the reference, :113,146).
    ``tempered=True``: allows ``temperature != 1`` (samples exp(-U/T); see mcmc/hmc.py here)."""

    def __init__(self, *a, trajectory_length=None, tempered=False, **kw):
        super().__init__(*a, **kw)
        assert trajectory_length is None or trajectory_length >= 1
        self.trajectory_length, self.tempered = trajectory_length, tempered
        self._since_mh = 0

    def _make_optimizer(self, params):
        # inference_reject.py:182-189
        assert self.tempered or self.temperature == 1.0, "HMC only implemented for temperature=1."
        assert self.momentum == 1.0, "HMC only works with momentum=1."
        assert self.descent_epochs == 0, "HMC not implemented for descent epochs with temp=0."
        extra = dict(temperature=self.temperature) if self.tempered else {}
        opt = mcmc.HMC(params=params, lr=self.learning_rate, num_data=self.eff_num_data,
                       **extra, **self._sampler_kwargs())
        # raise_on_nan keeps the reference's default (True, mcmc/hmc.py:25-27); inside a runner the test is the
        # device-side flag read at metric steps and epoch ends (SGLDRunner._check_finite), not a sync per step
        opt.defer_nan_check = True
        return opt

    def _trajectory_ends(self, last_of_epoch, epoch):
        if self.trajectory_length is None:
            return False
        self._since_mh += 1
        if last_of_epoch and self._is_sampling_epoch(epoch):
            # a SAMPLING epoch's end is an M-H point of its own (run_iter): the count restarts there.  Any other
            # epoch's end is an ordinary step boundary: the count runs on across it and may end exactly on it.
            return False
        return self._since_mh >= self.trajectory_length

    def _mh_point(self, step, acc, batches, save=None):
        self._since_mh = 0
        return super()._mh_point(step, acc, batches, save)


class SGLDRunnerReject(VerletSGLDRunnerReject):
    def _make_optimizer(self, params):
        # inference_reject.py:191-198
        assert not self.reject_samples
        return mcmc.SGLD(params=params, lr=self.learning_rate, num_data=self.eff_num_data,
                         momentum=self.momentum, temperature=self.temperature,
                         **self._sampler_kwargs())


# name -> runner class, as experiments/train_bnn.py:223-234 selects them
def runner_class(inference):
    from . import inference as plain
    table = {"SGLD": plain.SGLDRunner, "VerletSGLD": plain.VerletSGLDRunner,
             "OurHMC": plain.HMCRunner, "HMCReject": HMCRunnerReject,
             "VerletSGLDReject": VerletSGLDRunnerReject, "SGLDReject": SGLDRunnerReject}
    try:
        return table[inference]
    except KeyError:
        raise ValueError(f"Unknown inference method {inference}") from None
