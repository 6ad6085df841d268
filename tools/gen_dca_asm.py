#!/usr/bin/env python3
"""Writes helmnet_amd/csrc/hn_dca_pass.inc: the hand-scheduled gfx950 instruction streams of hn_dca.hip.

The conv1 block of one wavefront for ONE input channel of a staged chunk (hn_dca.hip, k_dc_asm): lane = mid column, the wave owns
R = 9 mid rows x all 8 mid channels (36 packed accumulators) plus the two edge columns of its rows on 18 lanes (4 packed
accumulators).  The 3x3 taps are walked kx-outer: pass kx reads ONE dword per input row (11 rows, ds_read_b32 with an immediate
offset) and keeps the 24 weights [ky][8 channels] of that kernel column in SGPRs (3 x s_load_dwordx8); per pass 108 + 12
v_pk_fma_f32.  Weights and row values of pass kx + 1 are requested at the top of pass kx (double-buffered SGPR / VGPR sets).

Operands of the asm statement (hn_dca.hip):
  %0 .. %35   acc[r][c]  (+v, 64-bit)   r = 0 .. 8 mid rows, c = 0 .. 3 channel pairs
  %36 .. %39  acce[c]    (+v, 64-bit)   edge position of this lane
  %40         v  byte address in LDS of (input row 9h, window column lane + 2) of this wave's channel plane
  %41         v  byte address in LDS of this lane's edge window (row 9h + (el >> 1), window column 66 + (el & 1))
  %42         s  64-bit address of this channel's weights [3 kx][3 ky][8] (fp32)
Fixed registers (clobbers): s[40:87] weights (two sets of 24), v[100:127] row values (two sets of 14).
"""
import os

PITCH_BYTES = 72 * 4
R = 9
WSET = (40, 64)      # first SGPR of weight set 0 / 1
XSET = (100, 114)    # first VGPR of row-value set 0 / 1 (11 main rows + 3 edge rows)


def bcast(v):
    """source operand + modifiers that broadcast the 32-bit VGPR v into both halves of a packed operand (src1)"""
    if v % 2 == 0:
        return f"v[{v}:{v + 1}]", "op_sel_hi:[1,0,1]"
    return f"v[{v - 1}:{v}]", "op_sel:[0,1,0]"


def loads(p, lines):
    ws, xs = WSET[p & 1], XSET[p & 1]
    for k in range(3):
        lines.append(f"s_load_dwordx8 s[{ws + 8 * k}:{ws + 8 * k + 7}], %42, {hex(p * 96 + 32 * k)}")
    for j in range(R + 2):
        lines.append(f"ds_read_b32 v{xs + j}, %40 offset:{j * PITCH_BYTES + 4 * p}")
    for ky in range(3):
        lines.append(f"ds_read_b32 v{xs + R + 2 + ky}, %41 offset:{ky * PITCH_BYTES + 4 * p}")


def fmas(p, lines):
    ws, xs = WSET[p & 1], XSET[p & 1]
    edge = []
    for ky in range(3):
        src, mod = bcast(xs + R + 2 + ky)
        for c in range(4):
            edge.append(f"v_pk_fma_f32 %{36 + c}, s[{ws + 8 * ky + 2 * c}:{ws + 8 * ky + 2 * c + 1}], {src}, %{36 + c} {mod}")
    for j in range(R + 2):
        src, mod = bcast(xs + j)
        for ky in range(3):
            r = j - ky
            if r < 0 or r >= R:
                continue
            for c in range(4):
                lines.append(f"v_pk_fma_f32 %{4 * r + c}, s[{ws + 8 * ky + 2 * c}:{ws + 8 * ky + 2 * c + 1}], {src}, %{4 * r + c} {mod}")
        if j in (3, 6, 9):   # the edge FMAs ride between the rows (their operands arrived with the pass's last loads)
            lines.extend(edge[4 * (j // 3 - 1):4 * (j // 3)])


def cin_block(with_loads=True, with_fmas=True):
    lines = []
    if with_loads:
        loads(0, lines)
    for p in range(3):
        if with_loads:
            lines.append("s_waitcnt lgkmcnt(0)")
            if p < 2:
                loads(p + 1, lines)
        if with_fmas:
            fmas(p, lines)
    return lines


def conv2_block(nch=8, rows=4, pitch_bytes=66 * 4, chan_bytes=18 * 66 * 4):
    """conv2 (8 -> 8) of one wavefront: `rows` output rows x all 8 output channels over the `nch` mid channels resident in LDS, 3 * nch passes chained --
    the loads of pass q + 1 (next kernel column, or the next channel's first) are requested at the top of pass q; no barriers inside.
    Operands: %0 .. %(4 rows - 1) acc2[r][c] (+v, 64-bit); %(4 rows) v byte address in LDS of (mid row of the wave's first output row - 1 + 1 ..., see
    hn_dca.hip) ; %(4 rows + 1) s 64-bit address of the weights [nch][3 kx][3 ky][8]."""
    na = 4 * rows
    xset = (100, 100 + rows + 2)
    lines = []

    def loads(q):
        cm, kx = divmod(q, 3)
        ws, xs = WSET[q & 1], xset[q & 1]
        for k in range(3):
            lines.append(f"s_load_dwordx8 s[{ws + 8 * k}:{ws + 8 * k + 7}], %{na + 1}, {hex(cm * 288 + kx * 96 + 32 * k)}")
        for j in range(rows + 2):
            lines.append(f"ds_read_b32 v{xs + j}, %{na} offset:{cm * chan_bytes + j * pitch_bytes + 4 * kx}")

    def fmas(q):
        ws, xs = WSET[q & 1], xset[q & 1]
        for j in range(rows + 2):
            src, mod = bcast(xs + j)
            for ky in range(3):
                r = j - ky
                if r < 0 or r >= rows:
                    continue
                for c in range(4):
                    lines.append(f"v_pk_fma_f32 %{4 * r + c}, s[{ws + 8 * ky + 2 * c}:{ws + 8 * ky + 2 * c + 1}], {src}, %{4 * r + c} {mod}")

    loads(0)
    for q in range(3 * nch):
        lines.append("s_waitcnt lgkmcnt(0)")
        if q + 1 < 3 * nch:
            loads(q + 1)
        fmas(q)
    return lines


def main():
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "helmnet_amd", "csrc", "hn_dca_pass.inc")
    body = cin_block()
    n_fma = sum(1 for l in body if l.startswith("v_pk_fma"))
    assert n_fma == 3 * (3 * R * 4 + 12), n_fma
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen_dca_asm.py -- do not edit.  conv1 of one wavefront over one staged input channel (hn_dca.hip).\n")
        f.write(f"// {n_fma} v_pk_fma_f32, {sum(1 for l in body if l.startswith('ds_read'))} ds_read_b32, "
                f"{sum(1 for l in body if l.startswith('s_load'))} s_load_dwordx8\n")
        for name, lines in (("HN_DCA_CIN_ASM", body), ("HN_DCA_CIN_ASM_NOFMA", cin_block(True, False)), ("HN_DCA_CIN_ASM_FMAONLY", cin_block(False, True))):
            if name != "HN_DCA_CIN_ASM":
                f.write("// timing-only ablation (wrong results by construction)\n")
            f.write(f"#define {name} \\\n")
            for l in lines:
                f.write(f'    "{l}\\n" \\\n')
            f.write('    ""\n')
        c2 = conv2_block()
        assert sum(1 for l in c2 if l.startswith("v_pk_fma")) == 8 * 9 * 4 * 4
        f.write(f"// conv2 (8 -> 8) of one wavefront, 4 output rows x 8 channels over the 8 LDS-resident mid channels: {sum(1 for l in c2 if l.startswith('v_pk_fma'))} "
                f"v_pk_fma_f32, {sum(1 for l in c2 if l.startswith('ds_read'))} ds_read_b32, {sum(1 for l in c2 if l.startswith('s_load'))} s_load_dwordx8\n")
        f.write("#define HN_DCA_CONV2_ASM \\\n")
        for l in c2:
            f.write(f'    "{l}\\n" \\\n')
        f.write('    ""\n')
        clob = [f"s{i}" for i in range(40, 88)] + [f"v{i}" for i in range(100, 128)]
        f.write("#define HN_DCA_CIN_CLOBBERS " + ", ".join(f'"{c}"' for c in clob) + ', "memory"\n')
    print(out, len(body), "instructions")


if __name__ == "__main__":
    main()
