"""GPU: the driver loops with the reference's names (Supervise / Mean_Teacher / CPS / HPFG; sup_ACDC.py:59-125,
2017_03_NIPS_Mean-Teacher_ACDC.py:63-162, 2021_06_CVPR_CPS_ACDC.py:61-169, main.py:79-289) end to end on build_loader("synthetic"):
iteration law (returns once cur_itrs > total_itrs), evaluation every step_size iterations through test_acdc, best-Dice checkpoints in the
reference's dict format, and resuming an optimizer from one."""
import os
from copy import deepcopy

import pytest
import torch

from hpfg_amd.datasets import build_loader
from hpfg_amd.model import build_model
from hpfg_amd.train import CPS, HPFG, Mean_Teacher, Supervise
from hpfg_amd.utils import AttrDict, build_optimizer

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class _Log:
    def __init__(self):
        self.lines = []

    def info(self, msg):
        self.lines.append(str(msg))


def _opt(**kw):
    d = dict(opt="sgd", lr=0.01, momentum=0.9, weight_decay=1e-4, sched="medical", warmup_epochs=0, warmup_lr=1e-4, min_lr=1e-6, total_itrs=4, step_size=2)
    d.update(kw)
    return AttrDict(d)


def _args(tmp, datasets="synthetic", **kw):
    a = _opt(datasets=datasets, model="unet", in_channels=1, num_classes=4, batch_size=2, unlabel_batch_size=2, train_crop_size=(64, 64),
             test_crop_size=(64, 64), synthetic_labeled=8, synthetic_unlabeled=12, synthetic_test_volumes=2, device=DEV, consistency=0.1,
             consistency_rampup=200.0, ema_decay=0.99, save_path=str(tmp), logger=_Log())
    a.update(kw)
    os.makedirs(os.path.join(str(tmp), "model"), exist_ok=True)
    for k in ("model", "ema_model", "model1", "model2"):
        a[f"{k}_save_path"] = os.path.join(str(tmp), "model", f"{k}.pth")
    return a


def _teacher(m):
    e = deepcopy(m)
    for p in e.parameters():
        p.requires_grad = False
    return e


def _check_ckpt(path, model, keys=("model", "optimizer", "lr_scheduler", "cur_itrs", "best_dice")):
    ck = torch.load(path, weights_only=False)
    assert set(keys) <= set(ck)
    assert set(ck["model"].keys()) == set(model.state_dict().keys())
    assert ck["cur_itrs"] in (2, 4) and 0.0 < ck["best_dice"] <= 1.0
    return ck


def test_supervise_loop(tmp_path):
    a = _args(tmp_path, datasets="sup_synthetic", weight_decay=5e-4, sched="cosine")
    torch.manual_seed(1)
    m = build_model(a).to(DEV)
    train_loader, test_loader = build_loader(a)
    log = Supervise(m, train_loader, test_loader, a)
    assert log.shape == (a.total_itrs + 1,) and torch.isfinite(log).all()          # the reference stops once cur_itrs > total_itrs
    files = os.listdir(os.path.join(str(tmp_path), "model"))
    assert any(f.startswith("model_0.") for f in files), files                      # sup_ACDC.py:114 names the file by its Dice
    assert any("dice" in ln for ln in a.logger.lines)
    assert m.training


def test_mean_teacher_loop_checkpoints_and_resume(tmp_path):
    a = _args(tmp_path)
    torch.manual_seed(1337)
    m = build_model(a).to(DEV)
    e = _teacher(m)
    lab, unl, test = build_loader(a)
    log = Mean_Teacher(m, e, lab, unl, test, a)
    assert log.shape == (a.total_itrs + 1,) and torch.isfinite(log).all()
    ck = _check_ckpt(a.model_save_path, m)
    _check_ckpt(a.ema_model_save_path, e)
    assert m.training and e.training
    # resume: a fresh optimizer takes the saved momentum (FusedSGD.state_dict carries the flat buffer)
    opt = build_optimizer(a, m)
    opt.load_state_dict(ck["optimizer"])
    assert float(opt._mom.abs().max()) > 0 and torch.equal(opt._mom, ck["optimizer"]["flat_momentum"].to(opt._mom.device))
    m2 = build_model(a).to(DEV)
    m2.load_state_dict(ck["model"])
    assert torch.equal(m2.state_dict()["decoder.out_conv.weight"], ck["model"]["decoder.out_conv.weight"].to(DEV))


def test_cps_loop(tmp_path):
    a = _args(tmp_path)
    a.model1, a.model2 = _opt(), _opt()
    torch.manual_seed(1337)
    m1, m2 = build_model(a).to(DEV), build_model(a).to(DEV)
    lab, unl, test = build_loader(a)
    log = CPS(m1, m2, lab, unl, test, a)
    assert log.shape == (a.total_itrs + 1,) and torch.isfinite(log).all()
    _check_ckpt(a.model1_save_path, m1)
    _check_ckpt(a.model2_save_path, m2)


def test_hpfg_loop_with_label_repeat(tmp_path):
    """main.py:79-289 on a 2+4 batch (the labelled batch of the second iterator is repeated Nu//Nl = 2 times, :142-143)."""
    a = _args(tmp_path, model="unet_plus", unlabel_batch_size=4, weight_decay=5e-4)
    a.model1, a.model2 = _opt(weight_decay=5e-4), _opt(weight_decay=5e-4)
    torch.manual_seed(1)
    m1, m2 = build_model(a).to(DEV), build_model(a).to(DEV)
    e = _teacher(m2)
    lab, unl, test = build_loader(a)
    log = HPFG(m1, m2, e, lab, unl, test, a)
    assert log.shape == (a.total_itrs + 1,) and torch.isfinite(log).all()
    for path, net in ((a.model1_save_path, m1), (a.model2_save_path, m2), (a.ema_model_save_path, e)):
        _check_ckpt(path, net)
    assert sum("_dice" in ln for ln in a.logger.lines) >= 6           # three networks evaluated at iterations 2 and 4


class _Writer:
    def __init__(self):
        self.rows = []

    def add_scalar(self, name, value, itr):
        self.rows.append((name, float(value), int(itr)))


def _mt_run(tmp, graph, writer=None):
    from hpfg_amd.model import reset_dropout_streams
    reset_dropout_streams()
    a = _args(tmp, total_itrs=5, step_size=1000, hipgraph=graph, log_every=4)
    if writer is not None:
        a.writer = writer
    torch.manual_seed(1337)
    m = build_model(a).to(DEV)
    e = _teacher(m)
    lab, unl, _ = build_loader(a)
    log = Mean_Teacher(m, e, lab, unl, None, a)
    torch.cuda.synchronize()
    return log.cpu(), m.flat_params.detach().cpu().clone(), e.flat_params.detach().cpu().clone()


def test_mean_teacher_loop_graphed_equals_eager_bitwise(tmp_path):
    """The drop-in loop captures its step at iteration 2 and replays it from then on (the path bench.py times): same losses, same student
    and teacher parameters, bit for bit, as the eager launches of the same six iterations -- dropout seeds, learning-rate law, EMA included."""
    w = _Writer()
    lg, pg, eg = _mt_run(tmp_path / "g", True, w)
    le, pe, ee = _mt_run(tmp_path / "e", False)
    assert lg.shape == (6,) and torch.equal(lg, le), (lg, le)
    assert torch.equal(pg, pe) and torch.equal(eg, ee)
    # the reference's per-iteration scalars (2017_03_NIPS_Mean-Teacher_ACDC.py:111-113), delivered every log_every iterations and at the end
    names = {n for n, _, _ in w.rows}
    assert names == {"mean_teacher/loss", "mean_teacher/lr", "mean_teacher/consistency_weight"}
    its = sorted({i for _, _, i in w.rows})
    assert its == [1, 2, 3, 4, 5, 6]
    loss = [v for n, v, i in sorted(w.rows, key=lambda r: r[2]) if n == "mean_teacher/loss"]
    assert max(abs(a - float(b)) for a, b in zip(loss, lg)) < 1e-6
    lrs = [v for n, v, i in sorted(w.rows, key=lambda r: r[2]) if n == "mean_teacher/lr"]
    assert 0 < lrs[2] < lrs[1] <= lrs[0] <= 0.01          # read after lr_scheduler.step(), as the reference does (:108-110)


def test_hpfg_loop_scalars(tmp_path):
    """main.py:216-222: the seven scalar names of the HPFG driver, and loss == loss_sup + loss_semi."""
    a = _args(tmp_path, model="unet_plus", unlabel_batch_size=4, weight_decay=5e-4, total_itrs=3, step_size=1000, log_every=2)
    a.model1, a.model2 = _opt(weight_decay=5e-4), _opt(weight_decay=5e-4)
    a.writer = _Writer()
    torch.manual_seed(1)
    m1, m2 = build_model(a).to(DEV), build_model(a).to(DEV)
    e = _teacher(m2)
    lab, unl, _ = build_loader(a)
    log = HPFG(m1, m2, e, lab, unl, None, a)
    assert log.shape == (4,) and torch.isfinite(log).all()
    names = {n for n, _, _ in a.writer.rows}
    assert names == {"HPFG/loss", "HPFG/loss_semi", "HPFG/loss_sup", "HPFG/lr1", "HPFG/lr2", "HPFG/consistency_weight_cps", "HPFG/consistency_weight_mt"}
    by = {}
    for n, v, i in a.writer.rows:
        by.setdefault(i, {})[n] = v
    for i, d in by.items():
        assert abs(d["HPFG/loss"] - d["HPFG/loss_sup"] - d["HPFG/loss_semi"]) < 1e-5 and d["HPFG/loss_sup"] > 0


def _ts_run(tmp, driver, graph, writer=None, seed_np=7, **kw):
    """One run of a teacher / student driver with its per-iteration draws seeded (numpy Beta factors for ICT, torch device noise for UAMT)."""
    import numpy as np
    from hpfg_amd.model import reset_dropout_streams
    reset_dropout_streams()
    a = _args(tmp, total_itrs=4, step_size=2, hipgraph=graph, log_every=3, ict_alpha=0.2, **kw)
    if writer is not None:
        a.writer = writer
    torch.manual_seed(1337)
    np.random.seed(seed_np)
    m = build_model(a).to(DEV)
    e = _teacher(m)
    lab, unl, test = build_loader(a)
    torch.cuda.manual_seed(99)          # the UAMT noise fields are draws of the device generator
    log = driver(m, e, lab, unl, test, a)
    torch.cuda.synchronize()
    return a, log.cpu(), m, e


@pytest.mark.parametrize("name", ["ICT_MedSeg", "Uncertainty_Aware"])
def test_teacher_student_driver_loops(tmp_path, name):
    """2022_02_ISBI_ICT-MedSeg_ACDC.py:65-190 / 2019_07_MICCAI_Uncertainty_Aware_ACDC.py:82-217 with the reference's function names and
    signature: iteration law, both networks evaluated and checkpointed at step_size, the reference's scalar names, and the captured loop
    (iteration 2 on) equal to the eager one on the same draws."""
    import hpfg_amd.train as T
    driver = getattr(T, name)
    w = _Writer()
    a, lg, m, e = _ts_run(tmp_path / "g", driver, True, w)
    assert lg.shape == (a.total_itrs + 1,) and torch.isfinite(lg).all()
    _check_ckpt(a.model_save_path, m)
    _check_ckpt(a.ema_model_save_path, e)
    assert m.training and e.training
    names = {n for n, _, _ in w.rows}
    want = {f"{name}/loss", f"{name}/lr", f"{name}/consistency_weight", f"{name}/consistency_loss"}
    if name == "Uncertainty_Aware":
        want.add(f"{name}/threshold")          # 2019_07...py:177
    assert names == want, names
    assert sorted({i for _, _, i in w.rows}) == [1, 2, 3, 4, 5]
    cons = [v for n, v, _ in w.rows if n.endswith("consistency_loss")]
    assert all(v >= 0 for v in cons) and (max(cons) > 0 or name == "Uncertainty_Aware")          # (UAMT: every pixel is above the entropy threshold at first)
    _, le, m2, e2 = _ts_run(tmp_path / "e", driver, False)
    assert torch.allclose(lg, le, rtol=0, atol=1e-6), (lg, le)
    assert torch.allclose(m.flat_params, m2.flat_params, rtol=0, atol=1e-6) and torch.allclose(e.flat_params, e2.flat_params, rtol=0, atol=1e-6)


def _hpfg_run(tmp, graph):
    import numpy as np
    from hpfg_amd.model import reset_dropout_streams
    reset_dropout_streams()
    a = _args(tmp, total_itrs=7, step_size=3, hipgraph=graph, log_every=4, model="unet_plus", unlabel_batch_size=4, weight_decay=5e-4)
    a.model1, a.model2 = _opt(weight_decay=5e-4, total_itrs=7), _opt(weight_decay=5e-4, total_itrs=7)
    torch.manual_seed(1)
    np.random.seed(3)
    m1, m2 = build_model(a).to(DEV), build_model(a).to(DEV)
    e = _teacher(m2)
    lab, unl, test = build_loader(a)
    log = HPFG(m1, m2, e, lab, unl, test, a)
    torch.cuda.synchronize()
    return log.cpu(), [m.flat_params.detach().cpu().clone() for m in (m1, m2, e)]


def test_hpfg_loop_graphed_equals_eager_bitwise(tmp_path):
    """main.py:79-289 with two evaluations of three networks in between (step_size 3): the captured loop == the eager loop, bit for bit --
    three U-Net+ engines, the projection necks, the dense loss, CutMix draws, per-forward dropout seeds of networks that run two forwards
    a step.  (Round 5: equal only since evaluation forwards stopped consuming seeds and the first conv's weight gradient became
    reproducible beside other streams' kernels -- csrc/common.h, HPFG_NO_PK_F32.)"""
    lg, pg = _hpfg_run(tmp_path / "g", True)
    le, pe = _hpfg_run(tmp_path / "e", False)
    assert lg.shape == (8,) and torch.equal(lg, le), (lg, le)
    assert all(torch.equal(a, b) for a, b in zip(pg, pe))
