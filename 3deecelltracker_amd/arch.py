"""Architecture tables for the three fixed-size 3D U-Nets and the FFN.

These are *data* (layer widths, pool sizes, activation kinds) restated from the
reference's Keras graph definitions so that the device engine, the oracle and the
synthetic-weight generator all walk the same layer list:

* unet3_a  -- reference CellTracker/unet3d.py:26-37 + _unet3_depth3 :84-98
* unet3_b  -- reference CellTracker/unet3d.py:40-67
* unet3_c  -- reference CellTracker/unet3d.py:70-81
* FFN      -- reference CellTracker/ffn.py:225-265

Keras semantics used everywhere (TF/Keras 2.11, third party): Conv3D is channels-last
cross-correlation with kernel layout (kx,ky,kz,Cin,Cout), padding='same' -> 1-voxel zero pad;
LeakyReLU() alpha = 0.3; BatchNormalization() eps = 1e-3 on the last axis using moving
statistics at inference; MaxPooling3D stride = pool; UpSampling3D = nearest repeat;
concatenate([up, skip]) puts the up-sampled channels first.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple

LEAKY_ALPHA = 0.3  # keras LeakyReLU() default
BN_EPS = 1e-3      # keras BatchNormalization() default

ACT_LEAKY = 0
ACT_RELU = 1


@dataclass(frozen=True)
class UNetArch:
    name: str
    arch_id: int
    input_shape: Tuple[int, int, int]          # (x, y, z) of one patch
    pool: Tuple[int, int, int]
    act: int                                   # ACT_LEAKY / ACT_RELU
    down: Tuple[Tuple[int, int], ...]          # (f1, f2) per down block
    up: Tuple[Tuple[int, int], ...]            # (f1, f2) per up block (deepest first)
    out: Tuple[int, int]                       # the two trailing conv widths

    def conv_layers(self) -> List[Tuple[int, int]]:
        """(Cin, Cout) of every 3x3x3 conv in execution order."""
        layers = []
        c = 1
        skips = []
        for f1, f2 in self.down:
            layers.append((c, f1)); layers.append((f1, f2))
            skips.append(f2); c = f2
        for (f1, f2), skip_c in zip(self.up, reversed(skips)):
            layers.append((c, f1)); layers.append((f1, f2))
            c = f2 + skip_c
        layers.append((c, self.out[0])); layers.append((self.out[0], self.out[1]))
        return layers

    def flops_per_patch(self) -> float:
        """2*MAC of all convs incl. the final 1x1x1 (35.57 GFLOP for unet3_a)."""
        x, y, z = self.input_shape
        px, py, pz = self.pool
        vox_levels = []
        for lvl in range(len(self.down) + 1):
            vox_levels.append((x // px ** lvl) * (y // py ** lvl) * (z // pz ** lvl))
        macs = 0
        convs = self.conv_layers()
        i = 0
        for lvl in range(len(self.down)):
            for _ in range(2):
                macs += vox_levels[lvl] * 27 * convs[i][0] * convs[i][1]; i += 1
        for k in range(len(self.up)):
            lvl = len(self.down) - k
            for _ in range(2):
                macs += vox_levels[lvl] * 27 * convs[i][0] * convs[i][1]; i += 1
        for _ in range(2):
            macs += vox_levels[0] * 27 * convs[i][0] * convs[i][1]; i += 1
        macs += vox_levels[0] * self.out[1]
        return 2.0 * macs

    def algorithmic_bytes_per_patch(self) -> float:
        """fp32 activation bytes if every conv reads its input once and writes its output once
        with pool / upsample / concat / sigmoid fused into the neighbouring conv (SURVEY 8d:
        285.1 MB for unet3_a) plus the packed weights."""
        x, y, z = self.input_shape
        px, py, pz = self.pool
        vox = [(x // px ** l) * (y // py ** l) * (z // pz ** l) for l in range(len(self.down) + 1)]
        convs = self.conv_layers()
        b = 0
        i = 0
        for lvl in range(len(self.down)):
            for _ in range(2):
                b += vox[lvl] * (convs[i][0] + convs[i][1]); i += 1
        for k in range(len(self.up)):
            lvl = len(self.down) - k
            for _ in range(2):
                b += vox[lvl] * (convs[i][0] + convs[i][1]); i += 1
        for _ in range(2):
            b += vox[0] * (convs[i][0] + convs[i][1]); i += 1
        b += vox[0] * (self.out[1] + 1)  # 1x1x1 sigmoid head: read out[1] channels, write one
        return 4.0 * b

    def weight_bytes(self) -> float:
        convs = self.conv_layers()
        return 4.0 * (sum(27 * ci * co + 3 * co for ci, co in convs) + self.out[1] + 1)


UNET3_A = UNetArch("unet3_a", 0, (160, 160, 16), (2, 2, 1), ACT_LEAKY,
                   ((8, 16), (16, 32), (32, 64)), ((64, 64), (32, 32), (16, 16)), (8, 8))
UNET3_B = UNetArch("unet3_b", 1, (96, 96, 8), (2, 2, 1), ACT_RELU,
                   ((64, 64), (128, 128)), ((256, 256), (128, 128)), (64, 64))
UNET3_C = UNetArch("unet3_c", 2, (64, 64, 64), (2, 2, 2), ACT_LEAKY,
                   ((8, 16), (16, 32), (32, 64)), ((64, 64), (32, 32), (16, 16)), (8, 8))

ARCHS = {a.name: a for a in (UNET3_A, UNET3_B, UNET3_C)}

# FFN (reference ffn.py:225-265)
FFN_K_PTRS = 20
FFN_FEAT = 61            # 20*3 relative coords + mean distance
FFN_HID = 512
