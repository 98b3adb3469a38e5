#!/usr/bin/env python3
"""Writes tests/golden/kats.json: the six hand-assembled HTJ2K known-answer codestreams of
SURVEY.md section 8(c) with the framecrc (Adler-32, seed 0, of the raw frame) and pixel
facts that the reference decoder (compiled during the survey session; it cannot be built in
later rounds) produced for them.  These are data: codestream bytes in, frame
checksum + a few pixel values out.  Nothing here is reference source text."""
import json
import os

KATS = [
    dict(name="KAT-1", what="one 64x64 HT block, 5/3, NL=0, sample(0,0) = -1",
         pix_fmt="gray", width=64, height=64, framecrc="0xb0690077",
         pixels={"0,0": 127}, others=128,
         hex="ff4fff510029400000000040000000400000000000000000000000400000004000000000000000000001070101"
             "ff500008000200000000ff52000c00000001000004044001ff5c00044040ff90000a00000000001b0001ff93"
             "c02a80017fff7fff7fff006900ffd9"),
    dict(name="KAT-2", what="u_off=1, u=3, emb_k=emb_1=1, 3 MagSgn bits: sample(0,0) = -5",
         pix_fmt="gray", width=64, height=64, framecrc="0x70690073",
         pixels={"0,0": 123}, others=128,
         hex="ff4fff510029400000000040000000400000000000000000000000400000004000000000000000000001070101"
             "ff500008000200000000ff52000c00000001000004044001ff5c00044040ff90000a00000000001c0001ff93"
             "c02ac0017fff7fff7fff01077a00ffd9"),
    dict(name="KAT-3", what="NL=1: sub-band order LL,HL,LH,HH, Mallat placement, per-band M_b/zbp, 5/3 IDWT",
         pix_fmt="gray", width=64, height=64, framecrc="0xc3240068",
         top_left_4x4=[[124, 125, 128, 128], [122, 128, 127, 127], [129, 127, 128, 128], [128, 127, 128, 128]],
         n_not_128=8,
         hex="ff4fff510029400000000040000000400000000000000000000000400000004000000000000000000001070101"
             "ff500008000200000000ff52000c00000001000104044001ff5c00074040484850ff90000a0000000000360001ff93"
             "c02a00017fff7f01077700c013c013c00a80007fff7f006600017fff7f006600007fff7f01077700ffd9"),
    dict(name="KAT-4", what="3 components, LRCP packet order, RCT, packed rgb24 store",
         pix_fmt="rgb24", width=64, height=64, framecrc="0xf2830159",
         pixels={"0,0": [122, 123, 124]}, others=128,
         hex="ff4fff51002f400000000040000000400000000000000000000000400000004000000000000000000003070101070101070101"
             "ff500008000200000000ff52000c00000001010004044001ff5c00044040ff90000a0000000000360001ff93"
             "c02ac0017fff7fff7fff01077a00c02a80007fff7fff7fff006900c02a80017fff7fff7fff006900ffd9"),
    dict(name="KAT-5", what="irreversible path: CAP HTIRV, QCD scalar expounded, step sizes, float dequant, 9/7 float IDWT, lrintf",
         pix_fmt="gray", width=64, height=64, framecrc="0xc11d0068",
         rows0_3_cols0_2=[[121, 126, 128], [118, 130, 126], [130, 127, 129], [129, 128, 128]],
         n_not_128=9,
         hex="ff4fff510029400000000040000000400000000000000000000000400000004000000000000000000001070101"
             "ff500008000200000020ff52000c00000001000104044000ff5c000b4240004a004c005600ff90000a0000000000370001ff93"
             "c02a00017fff7f01077700c0151004f002a0007fff7f01077700017fff7f006600007fff7f01077700ffd9"),
    dict(name="KAT-6", what="SigProp + MagRef passes and three-pass length signalling",
         pix_fmt="gray", width=64, height=64, framecrc="0x80a90074",
         pixels={"0,0": 125, "1,0": 127}, others=128,   # keys are "row,col"
         hex="ff4fff510029400000000040000000400000000000000000000000400000004000000000000000000001070101"
             "ff500008000200000000ff52000c00000001000004044001ff5c00044040ff90000a00000000001e0001ff93"
             "c072a100017fff7fff7fff0069002101ffd9"),
]

if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kats.json")
    with open(out, "w") as f:
        json.dump(KATS, f, indent=1)
    print("wrote", out)
