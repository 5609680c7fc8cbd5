"""Drop-in check of the Python plugin surface, operator by operator.

What a maintainer of the reference gets after switching the two import roots is a set of factories and callables with the reference's names:
``build_model``, ``build_optimizer``, ``build_lr_scheduler``, ``Med_Sup_Loss``, ``update_ema_variables``, ``get_current_consistency_weight``.
This test drives exactly those -- plus plain torch ops on their outputs (``softmax``, a mean of squared differences, ``Tensor.backward``) --
WITHOUT the fused step objects of hpfg_amd/train.py, through a small harness of this repository's own (``_Plugin`` below): one network
forward per call, one loss object call, one ``optimizer.step()``, one scheduler tick, one EMA call per iteration.  What is asserted are the
reference's own numbers: tests/golden/trace_mt.npz and trace_sup.npz hold the inputs, the dropout masks and the per-iteration losses /
final logits that oracle/make_golden.py recorded from the reference's modules (laws: 2017_03_NIPS_Mean-Teacher_ACDC.py:95-113 for the
student / teacher iteration, sup_ACDC.py:83-93 for the supervised one).
"""
import math
from copy import deepcopy
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import pytest
import torch

from hpfg_amd.model import build_model
from hpfg_amd.utils import (AttrDict, Med_Sup_Loss, build_lr_scheduler, build_optimizer, get_current_consistency_weight,
                            update_ema_variables)
from tests import trace_replay as R
from tests.helpers import maxerr
from tests.test_gpu_steps import _masks, _opt_args, logit_tol

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
TOL = 1e-3


def _cfg(**kw):
    c = _opt_args(**kw)
    c.update(dict(model="unet", in_channels=1, num_classes=4, device=DEV))
    return AttrDict(c)


@dataclass
class _Plugin:
    """The objects the plugin surface hands out for one run, and the handful of operator calls an iteration is made of."""
    cfg: AttrDict
    seed: int
    math_mode: str
    with_teacher: bool = False
    net: torch.nn.Module = field(init=False)
    teacher: Optional[torch.nn.Module] = field(init=False, default=None)
    history: List[list] = field(init=False, default_factory=list)

    def __post_init__(self):
        torch.manual_seed(self.seed)
        self.net = build_model(self.cfg).to(DEV)          # factory call #1 of the surface
        self.net.math = self.math_mode
        if self.with_teacher:
            self.teacher = deepcopy(self.net)            # the reference copies its student to get the EMA network
            for w in self.teacher.parameters():
                w.requires_grad = False
            self.teacher.train()
        self.net.train()
        self.opt = build_optimizer(args=self.cfg, model=self.net)           # factory call #2
        self.sched = build_lr_scheduler(args=self.cfg, optimizer=self.opt)  # factory call #3
        self.sup_loss = Med_Sup_Loss(self.cfg.num_classes)
        self.tick = 0

    def update(self, total: torch.Tensor):
        """zero_grad -> backward -> optimizer -> scheduler (-> EMA of the teacher), each through its public entry point."""
        self.opt.zero_grad()
        total.backward()
        self.opt.step()
        self.sched.step()
        if self.teacher is not None:
            update_ema_variables(self.net, self.teacher, self.cfg.ema_decay, self.tick)

    def lr(self) -> float:
        return float(self.opt.param_groups[0]["lr"])


def _student_teacher_iteration(run: _Plugin, fx, k: int):
    """Iteration k (1-based) of the Mean-Teacher law on the fixture's batch; returns (student logits, teacher logits, consistency weight)."""
    run.tick = k
    lab = torch.from_numpy(fx["xl"]).to(DEV)
    unl = torch.from_numpy(fx["xu"]).to(DEV)
    gt = torch.from_numpy(fx["yl"]).to(DEV).long()
    n_lab = lab.shape[0]
    run.net.external_dropout_masks = _masks(fx, f"it{k - 1}_s", 4, 32)          # the reference run's own nn.Dropout draws
    run.teacher.external_dropout_masks = _masks(fx, f"it{k - 1}_t", 4, 32)
    both = torch.cat((lab, unl))
    s_logits = run.net(both)
    with torch.no_grad():
        t_logits = run.teacher(both)
    s_prob, t_prob = s_logits.softmax(1), t_logits.softmax(1)
    sup = run.sup_loss(s_logits[:n_lab], gt)
    cons = ((s_prob[n_lab:] - t_prob[n_lab:]) ** 2).mean()
    w = get_current_consistency_weight(epoch=k // 150, args=run.cfg)
    total = sup + w * cons
    run.update(total)
    run.history.append([float(total), float(sup), float(cons)])
    return s_logits.detach(), t_logits, w


@pytest.mark.parametrize("math_mode", ["f32", "bf16x3"])
def test_plugin_surface_reproduces_the_reference_mean_teacher_trace(golden_dir, math_mode):
    fx = np.load(f"{golden_dir}/trace_mt.npz")
    # the fixture's consistency weight is the ramp-up law at epoch 0 for this `consistency` (exp(-5) * consistency)
    cfg = _cfg(consistency=float(fx["cons_w"]) / math.exp(-5.0), consistency_rampup=200.0)
    run = _Plugin(cfg, seed=1337, math_mode=math_mode, with_teacher=True)
    for k in (1, 2, 3):
        s_last, t_last, w = _student_teacher_iteration(run, fx, k)
    assert abs(w - float(fx["cons_w"])) < 1e-12 and run.lr() > 0
    assert np.abs(np.array(run.history) - fx["losses"]).max() < TOL, (run.history, fx["losses"])
    tol = logit_tol(math_mode, "mt", R.replay_mt, ["student_logits_last", "teacher_logits_last"])
    assert maxerr(s_last.cpu(), torch.from_numpy(fx["student_logits_last"])) < tol
    assert maxerr(t_last.cpu(), torch.from_numpy(fx["teacher_logits_last"])) < tol


@pytest.mark.parametrize("math_mode", ["f32", "bf16x3"])
def test_plugin_surface_reproduces_the_reference_supervised_trace(golden_dir, math_mode):
    fx = np.load(f"{golden_dir}/trace_sup.npz")
    run = _Plugin(_cfg(weight_decay=5e-4, sched="cosine"), seed=1, math_mode=math_mode)
    images = torch.from_numpy(fx["x"]).to(DEV)
    gt = torch.from_numpy(fx["labels"]).to(DEV).long()
    for k in range(1, 5):
        run.tick = k
        run.net.external_dropout_masks = _masks(fx, f"it{k - 1}_mask", 4, 32)
        value = run.sup_loss(run.net(images), gt)
        run.update(value)
        run.history.append(float(value))
    assert run.lr() > 0 and np.abs(np.array(run.history) - fx["losses"]).max() < TOL, (run.history, fx["losses"])
    run.net.eval()
    with torch.no_grad():
        final = run.net(images).cpu()
    assert maxerr(final, torch.from_numpy(fx["final_eval_logits"])) < logit_tol(math_mode, "sup", R.replay_sup, ["final_eval_logits"])
