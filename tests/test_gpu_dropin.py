"""Drop-in check of the Python plugin surface: the per-iteration statements of the reference's drivers --
2017_03_NIPS_Mean-Teacher_ACDC.py:82-113 and sup_ACDC.py:83-93 -- written out as test code, with only the import roots changed
(``hpfg_amd.utils`` / ``hpfg_amd.model`` instead of ``utils`` / ``model``).  Nothing here goes through the fused step objects of
hpfg_amd/train.py: ``model(x)``, ``torch.softmax``, ``Med_Sup_Loss``, ``torch.mean((a - b) ** 2)``, ``loss.backward()``,
``optimizer.step()``, ``lr_scheduler.step()`` and ``update_ema_variables`` are called one by one, as the drivers do.

Inputs, dropout masks and expected losses are the reference's own (tests/golden/trace_mt.npz, trace_sup.npz, written by
oracle/make_golden.py from the reference modules).  The only statements the reference does not have are the two that hand its dropout
masks to the networks (torch's CPU Philox stream cannot be re-drawn on the device) and the bookkeeping of the asserted values;
logging / TensorBoard / tqdm / evaluation lines are left out.
"""
import math
from copy import deepcopy

import numpy as np
import pytest
import torch

# ---- the two import lines a maintainer changes -------------------------------------------------------
from hpfg_amd.utils import get_current_consistency_weight, update_ema_variables, build_lr_scheduler, build_optimizer, Med_Sup_Loss
from hpfg_amd.model import build_model
# ----------------------------------------------------------------------------------------------------------------------------------
from hpfg_amd.utils import AttrDict
from tests.helpers import maxerr
from tests.test_gpu_steps import _masks, _opt_args, logit_tol
from tests import trace_replay as R

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
TOL = 1e-3


def _args(**kw):
    a = _opt_args(**kw)
    a.update(dict(model="unet", in_channels=1, num_classes=4, device=DEV))
    return AttrDict(a)


class _OneBatch:
    """A loader that yields the fixture's batch over and over (the reference iterates torch DataLoaders)."""

    def __init__(self, *tensors, n):
        self.tensors, self.n = tensors, n

    def __len__(self):
        return self.n

    def __iter__(self):
        for _ in range(self.n):
            yield self.tensors


@pytest.mark.parametrize("math_mode", ["f32", "bf16x3"])
def test_mean_teacher_loop_body_runs_unchanged(golden_dir, math_mode):
    d = np.load(f"{golden_dir}/trace_mt.npz")
    # the fixture was written with the consistency weight of cur_itrs // 150 == 40; with cur_itrs = 1..3 the driver's own law
    # (get_current_consistency_weight, ramp-up epoch 0) gives the same number for this `consistency`
    args = _args(consistency=float(d["cons_w"]) / math.exp(-5.0), consistency_rampup=200.0)
    torch.manual_seed(1337)
    model = build_model(args).to(args.device)
    model.math = math_mode
    ema_model = deepcopy(model)
    for p in ema_model.parameters():
        p.requires_grad = False
    label_loader = _OneBatch(torch.from_numpy(d["xl"]), torch.from_numpy(d["yl"]), n=3)
    unlabel_loader = _OneBatch(torch.from_numpy(d["xu"]), torch.zeros(2), n=3)
    rows = []

    # ---- 2017_03_NIPS_Mean-Teacher_ACDC.py:63-113 ----
    optimizer = build_optimizer(args=args, model=model)
    lr_scheduler = build_lr_scheduler(args=args, optimizer=optimizer)
    med_loss = Med_Sup_Loss(args.num_classes)
    model.train()
    ema_model.train()
    cur_itrs = 0
    label_iter = iter(label_loader)
    for epoch in range(1):
        train_loss = 0.0
        for i, (unlabel_img, _) in enumerate(unlabel_loader):
            cur_itrs += 1
            try:
                label_img, target_label = next(label_iter)
            except StopIteration:
                label_iter = iter(label_loader)
                label_img, target_label, = next(label_iter)

            label_img = label_img.to(args.device).float()
            unlabel_img = unlabel_img.to(args.device).float()
            target_label = target_label.to(args.device).long()
            label_bs = label_img.shape[0]
            model.external_dropout_masks = _masks(d, f"it{cur_itrs - 1}_s", 4, 32)          # (test only: the reference run's masks)
            ema_model.external_dropout_masks = _masks(d, f"it{cur_itrs - 1}_t", 4, 32)      # (test only)

            x = torch.cat([label_img, unlabel_img], dim=0)
            x = x.to(args.device).float()
            output = model(x)
            output_soft = torch.softmax(output, dim=1)

            with torch.no_grad():
                ema_output = ema_model(x)
                ema_output_soft = torch.softmax(ema_output, dim=1)

            loss_sup = med_loss(output[:label_bs], target_label)
            loss_consistence = torch.mean((output_soft[label_bs:] - ema_output_soft[label_bs:]) ** 2)
            consistency_weight = get_current_consistency_weight(epoch=cur_itrs // 150, args=args)
            loss = loss_sup + consistency_weight * loss_consistence
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
            lr_scheduler.step()
            train_loss += loss.item()
            lr = optimizer.param_groups[0]["lr"]
            update_ema_variables(model, ema_model, args.ema_decay, cur_itrs)
            rows.append([loss.item(), loss_sup.item(), loss_consistence.item()])          # (test only)
    # ---- end of the reference's statements ----

    assert abs(consistency_weight - float(d["cons_w"])) < 1e-12 and lr > 0
    assert np.abs(np.array(rows) - d["losses"]).max() < TOL, (rows, d["losses"])
    tol = logit_tol(math_mode, "mt", R.replay_mt, ["student_logits_last", "teacher_logits_last"])
    assert maxerr(output.detach().cpu(), torch.from_numpy(d["student_logits_last"])) < tol
    assert maxerr(ema_output.cpu(), torch.from_numpy(d["teacher_logits_last"])) < tol


@pytest.mark.parametrize("math_mode", ["f32", "bf16x3"])
def test_supervised_loop_body_runs_unchanged(golden_dir, math_mode):
    d = np.load(f"{golden_dir}/trace_sup.npz")
    args = _args(weight_decay=5e-4, sched="cosine")
    torch.manual_seed(1)
    model = build_model(args).to(args.device)
    model.math = math_mode
    train_loader = _OneBatch(torch.from_numpy(d["x"]), torch.from_numpy(d["labels"]), n=4)
    losses = []

    # ---- sup_ACDC.py:59-93 ----
    optimizer = build_optimizer(args=args, model=model)
    lr_scheduler = build_lr_scheduler(args=args, optimizer=optimizer)
    criterion = Med_Sup_Loss(args.num_classes)
    model.train()
    cur_itrs = 0
    train_loss = 0.0
    for epoch in range(1):
        for i, (img, label_true) in enumerate(train_loader):
            cur_itrs += 1
            img = img.to(args.device).float()
            label_true = label_true.to(args.device).long()
            model.external_dropout_masks = _masks(d, f"it{cur_itrs - 1}_mask", 4, 32)       # (test only)
            label_pred = model(img)
            loss = criterion(label_pred, label_true)

            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
            lr_scheduler.step()
            lr = optimizer.param_groups[0]["lr"]
            train_loss += loss.item()
            losses.append(loss.item())          # (test only)
    # ---- end of the reference's statements ----

    assert lr > 0 and np.abs(np.array(losses) - d["losses"]).max() < TOL, (losses, d["losses"])
    model.eval()
    with torch.no_grad():
        fin = model(img).cpu()
    assert maxerr(fin, torch.from_numpy(d["final_eval_logits"])) < logit_tol(math_mode, "sup", R.replay_sup, ["final_eval_logits"])
