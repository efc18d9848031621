"""CycleScheduler restated (reference scheduler.py:221-320): linear warm-up from lr_max/divider to
lr_max over `warmup_proportion` of the run, then cosine down to lr_max/divider/1e4.  Pure host
scalar math; works on anything with `.param_groups` (torch optimisers and FlatAdam alike).
The reference steps it BEFORE optimizer.step() (train_faceoff_perceptual.py:104-107)."""
from math import cos, pi


def anneal_linear(start, end, proportion):
    return start + proportion * (end - start)


def anneal_cos(start, end, proportion):
    return end + (start - end) / 2 * (cos(pi * proportion) + 1)


class _Phase:
    def __init__(self, start, end, n_iter, fn):
        self.start, self.end, self.n_iter, self.fn, self.n = start, end, n_iter, fn, 0

    def step(self):
        self.n += 1
        return self.fn(self.start, self.end, self.n / self.n_iter)

    @property
    def is_done(self):
        return self.n >= self.n_iter


class CycleScheduler:
    def __init__(self, optimizer, lr_max, n_iter, momentum=(0.95, 0.85), divider=25, warmup_proportion=0.3,
                 phase=("linear", "cos")):
        self.optimizer = optimizer
        phase1 = int(n_iter * warmup_proportion)
        phase2 = n_iter - phase1
        lr_min = lr_max / divider
        fns = {"linear": anneal_linear, "cos": anneal_cos}
        self.lr_phase = [_Phase(lr_min, lr_max, phase1, fns[phase[0]]), _Phase(lr_max, lr_min / 1e4, phase2, fns[phase[1]])]
        self.momentum = momentum
        self.momentum_phase = []
        if momentum is not None:
            m1, m2 = momentum
            self.momentum_phase = [_Phase(m1, m2, phase1, fns[phase[0]]), _Phase(m2, m1, phase2, fns[phase[1]])]
        self.phase = 0

    def step(self):
        lr = self.lr_phase[self.phase].step()
        momentum = self.momentum_phase[self.phase].step() if self.momentum is not None else None
        for group in self.optimizer.param_groups:
            group["lr"] = lr
            if momentum is not None:
                if "betas" in group:
                    group["betas"] = (momentum, group["betas"][1])
                else:
                    group["momentum"] = momentum
        if self.lr_phase[self.phase].is_done:
            self.phase += 1
        if self.phase >= len(self.lr_phase):
            for p in self.lr_phase + self.momentum_phase:
                p.n = 0
            self.phase = 0
        return lr, momentum
