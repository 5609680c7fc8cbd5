"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's training-time augmentation (datasets/utils.py:73-84
random_rot_flip / random_rotate, :99-117 RandomGenerator.__call__), with the random generators passed in explicitly so that a
test can replay the same draws through hpfg_amd.datasets.device_pool.  Never imported by the product path."""
from __future__ import annotations

import numpy as np
from scipy import ndimage
from scipy.ndimage import zoom


def random_rot_flip(image, label, np_rng):
    k = np_rng.randint(0, 4)
    image = np.rot90(image, k)
    label = np.rot90(label, k)
    axis = np_rng.randint(0, 2)
    image = np.flip(image, axis=axis).copy()
    label = np.flip(label, axis=axis).copy()
    return image, label


def random_rotate(image, label, np_rng):
    angle = np_rng.randint(-20, 20)
    image = ndimage.rotate(image, angle, order=0, reshape=False)
    label = ndimage.rotate(label, angle, order=0, reshape=False)
    return image, label


def random_generator(image, mask, output_size, py_rng, np_rng):
    """-> (float32 [1,H,W], uint8 [H,W]) exactly like RandomGenerator(output_size)(image, mask)."""
    if py_rng.random() > 0.5:
        image, mask = random_rot_flip(image, mask, np_rng)
    elif py_rng.random() > 0.5:
        image, mask = random_rotate(image, mask, np_rng)
    x, y = image.shape
    image = zoom(image, (output_size[0] / x, output_size[1] / y), order=0)
    mask = zoom(mask, (output_size[0] / x, output_size[1] / y), order=0)
    return image.astype(np.float32)[None], mask.astype(np.uint8)
