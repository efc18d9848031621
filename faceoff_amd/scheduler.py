"""Cyclic learning-rate / momentum schedule with the interface of the reference's `CycleScheduler`
(scheduler.py:251-320; used at train_faceoff_perceptual.py:194-201 with `momentum=None`,
`warmup_proportion=0.05`, and stepped BEFORE `optimizer.step()`, :104-107).

Restated as a closed form: one cycle of `n_iter` steps is two segments,

    warm-up   steps 1 .. n1            n1 = int(n_iter * warmup_proportion)     lr: lr_max/divider -> lr_max
    anneal    steps n1+1 .. n_iter                                              lr: lr_max -> lr_max/divider/1e4

and the value at step i of a segment of length n is curve(start, end, i / n) with curve "linear" or "cos" (half cosine).
Momentum, when given as (high, low), runs high -> low during warm-up and back during the anneal and is written to
`betas[0]` (Adam-style groups) or `momentum`.  After `n_iter` steps the cycle restarts.  Pure host scalar math; works
on anything with `.param_groups` (torch optimisers and trainer.FlatAdam alike)."""
from math import cos, pi

_CURVES = {
    "linear": lambda a, b, t: a + t * (b - a),
    "cos": lambda a, b, t: b + (a - b) / 2 * (cos(pi * t) + 1),
}


class CycleScheduler:
    def __init__(self, optimizer, lr_max, n_iter, momentum=(0.95, 0.85), divider=25, warmup_proportion=0.3,
                 phase=("linear", "cos")):
        self.optimizer = optimizer
        self.n_iter = n_iter
        self.momentum = momentum
        n1 = int(n_iter * warmup_proportion)
        lr_lo = lr_max / divider
        hi, lo = momentum if momentum is not None else (None, None)
        # (length, curve, lr start, lr end, momentum start, momentum end)
        self.segments = ((n1, _CURVES[phase[0]], lr_lo, lr_max, hi, lo),
                         (n_iter - n1, _CURVES[phase[1]], lr_max, lr_lo / 1e4, lo, hi))
        self.i = 0                      # steps taken in the current cycle

    def at(self, i):
        """(lr, momentum) of step i (1-based) of a cycle."""
        n, curve, lr0, lr1, m0, m1 = self.segments[0]
        if i > n:
            i -= n
            n, curve, lr0, lr1, m0, m1 = self.segments[1]
        t = i / n
        return curve(lr0, lr1, t), (curve(m0, m1, t) if m0 is not None else None)

    def step(self):
        self.i += 1
        lr, mom = self.at(self.i)
        for group in self.optimizer.param_groups:
            group["lr"] = lr
            if mom is None:
                continue
            if "betas" in group:
                group["betas"] = (mom, group["betas"][1])
            else:
                group["momentum"] = mom
        if self.i >= self.n_iter:
            self.i = 0
        return lr, mom

    def state_dict(self):
        return {"i": self.i}

    def load_state_dict(self, sd):
        self.i = int(sd["i"])
