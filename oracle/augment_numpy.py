"""CPU oracle (TEST INFRASTRUCTURE ONLY) for the loader-side transforms of the reference (R.md:211-234).

torchvision / PIL are not installed here and the reference's transforms are unseeded, so this restates the documented
semantics with explicit random draws: RandomHorizontalFlip (flip first), RandomCrop(size, padding=4) (zero pad 4, crop at
(top, left) in 0..8), ToTensor (uint8 HWC -> float CHW / 255), AddGaussianNoise (x + randn * std, R.md:216-218).
Parity of the draws themselves is unpinned (the reference does not seed them); parity of the arithmetic is exact.
"""
import numpy as np


def augment_ref(u8, flips, tops, lefts, noise, std=0.03):
    b, h, w, _ = u8.shape
    out = np.zeros((b, 3, h, w), np.float32)
    for n in range(b):
        img = u8[n][:, ::-1, :] if flips[n] else u8[n]
        pad = np.zeros((h + 8, w + 8, 3), np.uint8)
        pad[4:4 + h, 4:4 + w] = img
        crop = pad[tops[n]:tops[n] + h, lefts[n]:lefts[n] + w]
        out[n] = crop.transpose(2, 0, 1).astype(np.float32) / np.float32(255.0)
    if noise is not None:
        out = out + np.float32(std) * noise.astype(np.float32)
    return out.astype(np.float32)
