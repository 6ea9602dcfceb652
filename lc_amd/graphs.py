"""hipGraph replay of the launch-bound pieces around the kernels (MI355X guideline: capture launch-bound inner loops).

A `Loss_fn` step is 25-75 short launches forward + backward for < 0.25 ms of kernel time, i.e. bound by the host's launch
rate.  `GraphedLoss` captures it with `torch.cuda.make_graphed_callables` (one forward and one backward graph per
configuration) and replays it: sparse heads, B=256 N=64 on one MI355X: 338 us eager -> 103 us graphed, loss values and
gradients bit-identical (`scripts/ubench/graph_lossfn.py`).  The test-time counterpart is `lc_amd.inference.GraphedSolvePnP`.

What makes the step capturable: no host synchronisation anywhere in the path, the `NormClipper` hooks keep their running
maximum at a fixed address (updated in place, `lc_amd/grad.py`), and the two host-side sources of variation are handled
outside the graph -- the dense heads' random sub-sampling phase (`np.random.randint`, `losses.py:152`) is drawn per call and
selects one of `dense_sample**2` lazily captured graphs; the warm-up blending factor (`losses.py:272-276`, a Python float) is
baked into a captured graph, so graphs are keyed by it: pass `step` to `__call__` and a call whose factor differs from the
captured one is never replayed with the stale blend -- inside the ramp (0 < factor < 1, a different value every step) the
step runs eagerly, at the plateaus (factor 0 or 1) a graph for that value is captured once.
"""
from __future__ import annotations

import numpy as np
import torch
from torch import Tensor

from .grad import NormClipper
from .losses import pose_loss_factor
from .inference import GraphedSolvePnP, quiet_capture  # noqa: F401  (GraphedSolvePnP is re-exported)


class GraphedLoss:
    """`loss_fn(gt_dict, out_dict, epoch, step, steps_per_epoch)` replayed as hipGraphs for the shapes of the example dicts.

        graphed = GraphedLoss(loss_fn, gt_dict, out_dict, epoch, step, steps_per_epoch)
        loss_dict, w_loss_dict = graphed(gt_dict, out_dict)          # same return value as loss_fn(...)
        sum(w_loss_dict.values()).backward()                         # backward replays the captured backward graph

    Every tensor entry of `out_dict` is a differentiable input, every tensor entry of `gt_dict` a constant input; non-tensor
    entries (bit counts, ...) are frozen at capture.  Sparse heads, continuous-xyz and binary-code dense heads are supported: the
    module's state (`NormClipper.max_norm`, the code histogram of `Loss_xyz_bin`) lives at fixed addresses and is updated in place.
    """

    def __init__(self, loss_fn, gt_dict: dict, out_dict: dict, epoch: int, step: int, steps_per_epoch: int):
        self.loss_fn = loss_fn
        self.when = (epoch, step, steps_per_epoch)
        self.out_keys = [k for k, v in out_dict.items() if isinstance(v, Tensor)]
        self.gt_keys = [k for k, v in gt_dict.items() if isinstance(v, Tensor)]
        self._frozen_out = {k: v for k, v in out_dict.items() if not isinstance(v, Tensor)}
        self._frozen_gt = {k: v for k, v in gt_dict.items() if not isinstance(v, Tensor)}
        self.dense = "pts2d" not in out_dict
        self.sample = int(loss_fn.cfg.pose_loss_cfg.get("dense_sample", 2)) if self.dense else 1
        self._example = ([out_dict[k].detach().clone().requires_grad_(out_dict[k].is_floating_point()) for k in self.out_keys] +
                         [gt_dict[k].detach().clone() for k in self.gt_keys])
        self._graphs = {}
        self._loss_keys = self._w_keys = None

    def _factor(self, when):
        cfg = self.loss_fn.cfg
        if not self.dense and not cfg.get("w_loss_pose", 0) > 0:
            return 1.0  # sparse heads without a pose term: nothing is blended
        return float(pose_loss_factor(cfg, when[1], when[2]))

    def _run(self, phase, *flat):
        n = len(self.out_keys)
        # Loss_fn hangs its gradient-clipping hooks on the network outputs it is given; the static placeholders of the capture
        # are reused by every warm-up pass, so hand it fresh aliases (a view: no launch) or the hooks would pile up on them
        flat = [t.view_as(t) if t.requires_grad else t for t in flat]
        out = dict(self._frozen_out, **dict(zip(self.out_keys, flat[:n])))
        gt = dict(self._frozen_gt, **dict(zip(self.gt_keys, flat[n:])))
        self.loss_fn._forced_phase = phase
        try:
            loss_dict, w_loss_dict = self.loss_fn(gt, out, *self.when)
        finally:
            self.loss_fn._forced_phase = None
        if self._loss_keys is None:
            self._loss_keys, self._w_keys = list(loss_dict), list(w_loss_dict)
        return tuple(loss_dict[k] for k in self._loss_keys) + tuple(w_loss_dict[k] for k in self._w_keys)

    def _capture(self, phase):
        # the warm-up iterations of make_graphed_callables run the step for real: keep them out of the module's state
        # (the clippers' running maxima, the code histogram), which must also sit on the device BEFORE the capture
        dev = self._example[0].device
        self.loss_fn.to(dev)
        for c in self.loss_fn.modules():
            if isinstance(c, NormClipper):
                c._ws(dev)
        buffers = [b for _, b in self.loss_fn.named_buffers()]
        saved = [b.detach().clone() for b in buffers]
        with quiet_capture():
            graphed = torch.cuda.make_graphed_callables(lambda *flat: self._run(phase, *flat), tuple(self._example))
        for b, s in zip(buffers, saved):
            b.copy_(s)
        return graphed

    def __call__(self, gt_dict: dict, out_dict: dict, epoch: int = None, step: int = None, steps_per_epoch: int = None):
        """Replays the step.  `step` (and optionally epoch / steps_per_epoch) are the arguments `loss_fn(...)` would get now: when
        the warm-up factor they imply is not the captured one, the stale graph is not used (see the module docstring)."""
        if step is not None:
            when = (self.when[0] if epoch is None else epoch, step, self.when[2] if steps_per_epoch is None else steps_per_epoch)
            factor = self._factor(when)
            if 0.0 < factor < 1.0:  # inside the ramp: a new blend every step -> run the step itself
                return self.loss_fn(gt_dict, out_dict, *when)
            if factor != self._factor(self.when):
                self._graphs.clear()  # plateau reached (or left): capture for the new constant blend
            self.when = when
        phase = tuple(int(v) for v in np.random.randint(0, self.sample, size=2)) if self.dense else None  # losses.py:152
        if phase not in self._graphs:
            self._graphs[phase] = self._capture(phase)
        flat = [out_dict[k] for k in self.out_keys] + [gt_dict[k] for k in self.gt_keys]
        res = self._graphs[phase](*flat)
        n = len(self._loss_keys)
        return dict(zip(self._loss_keys, res[:n])), dict(zip(self._w_keys, res[n:]))

