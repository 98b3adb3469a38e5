"""Catalogue of generated HTJ2K test streams shared by the CPU and GPU suites.
Each entry: name -> (callable returning the codestream bytes, decode kwargs)."""
import functools

import numpy as np

import vecgen


@functools.lru_cache(maxsize=None)
def _img(w, h, nc, depth=8, seed=1, noise=8, dx=None, dy=None):
    return vecgen.synth_image(w, h, nc, depth=depth, seed=seed, noise=noise,
                              dx=list(dx) if dx else None, dy=list(dy) if dy else None)


def _enc(img_args, **kw):
    return vecgen.encode(_img(*img_args), **kw)


CASES = {
    # --- reversible 5/3 ---
    "gray_l5_cb64":      (lambda: _enc((200, 150, 1, 8, 3)), {}),
    "gray_l5_cb32":      (lambda: _enc((200, 150, 1, 8, 3), cb=(5, 5)), {}),
    "gray_l3_cb16x64":   (lambda: _enc((200, 150, 1, 8, 3), cb=(4, 6), nlevels=3), {}),
    "gray_l3_cb256x16":  (lambda: _enc((300, 90, 1, 8, 13), cb=(8, 4), nlevels=3), {}),
    "gray_l2_cb4x1024":  (lambda: _enc((40, 1100, 1, 8, 14), cb=(2, 10), nlevels=2), {}),
    "gray_l0":           (lambda: _enc((200, 150, 1, 8, 3), nlevels=0), {}),
    "gray_offset":       (lambda: _enc((201, 149, 1, 8, 4), nlevels=2, offset=(3, 5)), {}),
    "gray_deep_levels":  (lambda: _enc((37, 23, 1, 8, 15), nlevels=8), {}),
    "rgb_mct":           (lambda: _enc((190, 131, 3, 8, 5), mct=1), {}),
    "rgb_nomct_rlcp":    (lambda: _enc((190, 131, 3, 8, 5), prog=1, nlevels=4), {}),
    "rgb_rpcl_prec":     (lambda: _enc((190, 131, 3, 8, 5), mct=1, prog=2, prec=[(7, 7), (6, 6)], nlevels=3), {}),
    "rgb_pcrl_prec":     (lambda: _enc((190, 131, 3, 8, 5), mct=1, prog=3, prec=[(7, 7), (6, 6)], nlevels=3), {}),
    "rgb_cprl_prec":     (lambda: _enc((190, 131, 3, 8, 5), mct=1, prog=4, prec=[(7, 7), (6, 6)], nlevels=3), {}),
    "rgb_tiles":         (lambda: _enc((190, 131, 3, 8, 5), mct=1, tile=(64, 64), nlevels=3), {}),
    "rgb_tiles_offsets": (lambda: _enc((190, 131, 3, 8, 6), mct=1, tile=(100, 70), nlevels=3, offset=(7, 9), tile_offset=(2, 3)), {}),
    "gray_sop_eph":      (lambda: _enc((200, 150, 1, 8, 3), sop=True, eph=True), {}),
    "gray12":            (lambda: _enc((160, 120, 1, 12, 7, 40), depth=12, nlevels=4), {}),
    "gray16":            (lambda: _enc((160, 120, 1, 16, 8, 400), depth=16, nlevels=4), {}),
    "rgb10_mct":         (lambda: _enc((160, 120, 3, 10, 8, 20), depth=10, nlevels=4, mct=1), {}),
    "rgba8":             (lambda: _enc((96, 80, 4, 8, 16), nlevels=3), {}),
    "yuv420p8":          (lambda: _enc((190, 130, 3, 8, 12, 8, (1, 2, 2), (1, 2, 2)), dx=[1, 2, 2], dy=[1, 2, 2], width=190, height=130), {}),
    "tiny_7x5":          (lambda: _enc((7, 5, 1, 8, 9), nlevels=1), {}),
    "tiny_1x1":          (lambda: vecgen.encode([np.array([[77]])], nlevels=0), {}),
    "tiny_3x1_l2":       (lambda: vecgen.encode([np.array([[77, 3, 250]])], nlevels=2), {}),
    "tiny_1x9_l3":       (lambda: vecgen.encode([np.arange(9).reshape(9, 1) * 20], nlevels=3), {}),
    "noise_max":         (lambda: vecgen.encode([np.random.default_rng(2).integers(0, 256, (130, 130))], nlevels=3), {}),
    "all_zero":          (lambda: vecgen.encode([np.full((100, 100), 128)], nlevels=3), {}),
    "one_sample_blocks": (lambda: vecgen.encode([np.where(np.add.outer(np.arange(128) % 64, np.arange(128) % 64) == 0, 200, 128)], nlevels=1), {}),
    "force_include":     (lambda: _enc((100, 100, 1, 8, 17), nlevels=3, force_include=True), {}),
    "psot_zero":         (lambda: _enc((100, 100, 1, 8, 17), nlevels=3, psot_zero=True), {}),
    "placeholder_1":     (lambda: _enc((200, 150, 1, 8, 3), placeholder_sets=1), {}),
    "placeholder_2_3p":  (lambda: _enc((200, 150, 1, 8, 3), placeholder_sets=2, passes=3), {}),
    "lowres_1":          (lambda: _enc((200, 150, 1, 8, 3)), {"reduction_factor": 1}),
    "lowres_3_rgb":      (lambda: _enc((190, 131, 3, 8, 5), mct=1), {"reduction_factor": 3}),
    # --- refinement passes ---
    "gray_3passes":      (lambda: _enc((200, 150, 1, 8, 3), passes=3), {}),
    "gray_2passes":      (lambda: _enc((200, 150, 1, 8, 3), passes=2), {}),
    "gray_3passes_vsc":  (lambda: _enc((200, 150, 1, 8, 3), passes=3, vsc=True), {}),
    "rgb_3passes_cb32":  (lambda: _enc((190, 131, 3, 8, 5), mct=1, passes=3, cb=(5, 5)), {}),
    # --- irreversible 9/7 (float path; bitexact=1 selects the fixed-point path) ---
    "gray_97_q2":        (lambda: _enc((200, 150, 1, 8, 3), transform=0, qstep=2), {}),
    "gray_97_fine":      (lambda: _enc((200, 150, 1, 8, 3), transform=0, qstep=1 / 32), {}),
    "rgb_97_ict":        (lambda: _enc((190, 131, 3, 8, 5), transform=0, mct=1, qstep=1), {}),
    "gray_97_3passes":   (lambda: _enc((200, 150, 1, 8, 3), passes=3, transform=0, qstep=2), {}),
    "yuv422p12_97":      (lambda: _enc((192, 128, 3, 12, 11, 30, (1, 2, 2), (1, 1, 1)), depth=12, dx=[1, 2, 2], dy=[1, 1, 1],
                                       transform=0, qstep=1, cb=(5, 5), width=192, height=128), {}),
    "gray_97_offset":    (lambda: _enc((201, 149, 1, 8, 4), nlevels=3, offset=(3, 5), transform=0, qstep=1), {}),
    "gray_97_bitexact":  (lambda: _enc((200, 150, 1, 8, 3), transform=0, qstep=2), {"bitexact": 1}),
    "rgb_97_bitexact":   (lambda: _enc((190, 131, 3, 8, 5), transform=0, mct=1, qstep=1), {"bitexact": 1}),
    # --- regressions found by tools/gpu_random_configs.py ---
    # 9/7 fixed point through the general path of the fused final level, last quadruple of a row partly outside the
    # line (width = 3 mod 4): a lane without output used to leave the lifting of its neighbour short of a DPP source
    "rgb_97_bitexact_w87": (lambda: _enc((87, 96, 3, 8, 365, 4), transform=0, qstep=4.0, mct=1, nlevels=2, cb=(6, 2)), {"bitexact": 1}),
    "rgb12_97_bitexact_w91": (lambda: _enc((91, 40, 3, 12, 366, 4), depth=12, transform=0, qstep=4.0, mct=1, nlevels=5, cb=(6, 2)), {"bitexact": 1}),
    # image offset + 4:2:0: the last chroma row of the picture is covered by no tile-component (jpeg2000dec.c:2312-2358)
    "yuv420_offset_uncovered_row": (lambda: _enc((377, 15, 3, 8, 169, 4, (1, 2, 2), (1, 2, 2)), nlevels=4, cb=(2, 6), offset=(2, 5), dx=[1, 2, 2], dy=[1, 2, 2], width=377, height=15), {}),
    # 42 tiles x 3 components = 126 tile-components per frame (a batch of three used to exceed a 250 limit)
    "yuv420_42_tiles":   (lambda: _enc((212, 168, 3, 8, 337, 4, (1, 2, 2), (1, 2, 2)), nlevels=2, cb=(6, 5), tile=(32, 32), prog=4, dx=[1, 2, 2], dy=[1, 2, 2], width=212, height=168), {}),
    # --- palettised JP2 (pclr + cmap boxes): pal8, the palette is AVFrame.data[1] (jpeg2000dec.c:2900-2901) ---
    "pal8_jp2":           (lambda: vecgen.jp2_wrap(vecgen.encode([(np.add.outer(np.arange(40), np.arange(56)) * 3 % 200).astype(np.int32)], nlevels=2),
                                                   56, 40, 1, 8, colourspace=16,
                                                   palette=[((i * 7) & 255, (255 - i) & 255, (i * 3 + 1) & 255) for i in range(200)]), {}),
    # --- Part-1 (MQ-coded) blocks: decode_cblk() instead of the HT decoder; style bits BYPASS 1, RESET 2,
    #     TERMALL 4, VSC 8, SEGSYM 0x20 ---
    "p1_gray":            (lambda: _enc((200, 150, 1, 8, 3), part1=True), {}),
    "p1_gray_cb32":       (lambda: _enc((200, 150, 1, 8, 3), part1=True, cb=(5, 5)), {}),
    "p1_gray_cb16x64":    (lambda: _enc((200, 150, 1, 8, 3), part1=True, cb=(4, 6), nlevels=3), {}),
    "p1_gray_cb64x4":     (lambda: _enc((200, 150, 1, 8, 3), part1=True, cb=(6, 2), nlevels=2), {}),
    "p1_gray_cb4x1024":   (lambda: _enc((40, 1100, 1, 8, 14), part1=True, cb=(2, 10), nlevels=2), {}),
    "p1_gray_cb128x32":   (lambda: _enc((300, 90, 1, 8, 13), part1=True, cb=(7, 5), nlevels=2), {}),
    "p1_cb256x16_modes":  (lambda: _enc((600, 70, 1, 12, 13, 60), depth=12, part1=True, cb=(8, 4), nlevels=1, cblk_style=0x29), {}),
    "p1_gray_cb1024x4":   (lambda: _enc((1100, 40, 1, 8, 14), part1=True, cb=(10, 2), nlevels=1), {}),
    "p1_bypass":          (lambda: _enc((200, 150, 1, 12, 3, 60), depth=12, part1=True, cblk_style=0x01), {}),
    "p1_reset":           (lambda: _enc((200, 150, 1, 8, 3), part1=True, cblk_style=0x02), {}),
    "p1_termall":         (lambda: _enc((200, 150, 1, 8, 3), part1=True, cblk_style=0x04), {}),
    "p1_vsc":             (lambda: _enc((200, 150, 1, 8, 3), part1=True, cblk_style=0x08), {}),
    "p1_segsym":          (lambda: _enc((200, 150, 1, 8, 3), part1=True, cblk_style=0x20), {}),
    "p1_bypass_termall":  (lambda: _enc((200, 150, 1, 12, 3, 60), depth=12, part1=True, cblk_style=0x05), {}),
    "p1_all_switches":    (lambda: _enc((160, 120, 1, 16, 8, 400), depth=16, nlevels=4, part1=True, cblk_style=0x2F), {}),
    "p1_rgb_mct":         (lambda: _enc((190, 131, 3, 8, 5), mct=1, part1=True), {}),
    "p1_rgb_tiles":       (lambda: _enc((190, 131, 3, 8, 6), mct=1, tile=(100, 70), nlevels=3, offset=(7, 9), tile_offset=(2, 3), part1=True), {}),
    "p1_yuv420":          (lambda: _enc((190, 130, 3, 8, 12, 8, (1, 2, 2), (1, 2, 2)), dx=[1, 2, 2], dy=[1, 2, 2], width=190, height=130, part1=True), {}),
    "p1_truncated_1":     (lambda: _enc((200, 150, 1, 8, 3), part1=True, drop_passes=1), {}),
    "p1_truncated_2":     (lambda: _enc((200, 150, 1, 8, 3), part1=True, drop_passes=2, cblk_style=0x04), {}),
    "p1_truncated_5":     (lambda: _enc((200, 150, 1, 8, 3), part1=True, drop_passes=5), {}),
    "p1_97":              (lambda: _enc((200, 150, 1, 8, 3), part1=True, transform=0, qstep=1), {}),
    "p1_97_rgb_bitexact": (lambda: _enc((190, 131, 3, 8, 5), part1=True, transform=0, mct=1, qstep=1), {"bitexact": 1}),
    "p1_lowres_2":        (lambda: _enc((190, 131, 3, 8, 5), mct=1, part1=True), {"reduction_factor": 2}),
    "p1_noise_max":       (lambda: vecgen.encode([np.random.default_rng(2).integers(0, 256, (130, 130))], nlevels=3, part1=True), {}),
    "p1_all_zero":        (lambda: vecgen.encode([np.full((100, 100), 128)], nlevels=3, part1=True), {}),
    "p1_tiny_3x1":        (lambda: vecgen.encode([np.array([[77, 3, 250]])], nlevels=2, part1=True), {}),
    # --- MIXED (SPcod bits 6-7 = 3): HT and Part-1 blocks in one stream, told apart in the packet header
    #     (jpeg2000dec.c:1256-1340).  OpenJPEG does not read MIXED streams: restatement + own round trip only ---
    "mixed_gray":         (lambda: _enc((200, 150, 1, 8, 3), mixed=True), {}),
    "mixed_rgb_cb32":     (lambda: _enc((190, 131, 3, 8, 5), mct=1, mixed=True, cb=(5, 5), nlevels=3), {}),
    "mixed_3passes_vsc":  (lambda: _enc((200, 150, 1, 8, 3), mixed=True, passes=3, vsc=True), {}),
    "mixed_gray16_tiles": (lambda: _enc((160, 120, 1, 16, 8, 400), depth=16, nlevels=3, mixed=True, tile=(96, 64)), {}),
}


@functools.lru_cache(maxsize=None)
def get(name):
    fn, kw = CASES[name]
    return fn(), kw
