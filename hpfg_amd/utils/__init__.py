"""Flat re-export of the hot-path utilities, as the reference's drivers import them (``from utils import ...``, main.py:10-12)."""
from .loss import DiceLoss, Med_Sup_Loss, Dense_Loss, seg_loss
from .optim import FusedAdamW, FusedSGD, build_lr_scheduler, build_optimizer, flatten_parameters
from .scheduler import CosineWarmupLR_Scheduler, Medical_LR, PolyLR
from .utils import (AttrDict, BoxMaskGenerator, ema_alpha, get_current_consistency_weight, linear_rampup, loadyaml, mk_path,
                    sigmoid_rampup, update_ema_variables, update_ema_variables_backbone)
from .logger import _get_logger
