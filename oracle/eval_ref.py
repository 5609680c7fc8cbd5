"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's evaluation (val.py:268-287 test_single_volume, :376-387
calculate_metric_percase with medpy 0.4.0 `binary.dc`), used to check hpfg_amd.val.  Never imported by the product path.

Per slice: zoom(order=0) to patch_size -> eval-mode forward (oracle U-Net, oracle/unet_ref.py) -> argmax(softmax) -> zoom(order=0)
back; per foreground class: dice = medpy dc if the class is predicted at all, else 0 (the reference's `elif` is unreachable)."""
from __future__ import annotations

import numpy as np
import torch
from scipy.ndimage import zoom

from . import losses_ref, unet_ref


def test_single_volume(image: np.ndarray, label: np.ndarray, state: dict, classes: int, patch_size=(256, 256)):
    """image, label: [S,h,w] numpy; state: oracle U-Net state (unet_ref.init_state layout).  Returns per-class dice list."""
    prediction = np.zeros_like(label)
    for ind in range(image.shape[0]):
        sl = image[ind]
        x, y = sl.shape
        sl = zoom(sl, (patch_size[0] / x, patch_size[1] / y), order=0)
        inp = torch.from_numpy(np.ascontiguousarray(sl)).unsqueeze(0).unsqueeze(0).float()
        with torch.no_grad():
            logits = unet_ref.unet_forward(state, inp, train=False)
            out = torch.argmax(torch.softmax(logits, dim=1), dim=1).squeeze(0).numpy()
        prediction[ind] = zoom(out, (x / patch_size[0], y / patch_size[1]), order=0)
    dices = []
    for c in range(1, classes):
        p, g = prediction == c, label == c
        dices.append(losses_ref.binary_dice(p, g) if p.sum() > 0 else 0.0)
    return dices, prediction


test_single_volume.__test__ = False
