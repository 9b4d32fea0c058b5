#!/usr/bin/env python
"""Per kernel family of the ViT-S img+txt client step, from ONE rocprofv3 --kernel-trace --stats run of bench.py: in-step launches per step
and their average duration, the algorithmic FLOPs and bytes of the family per step (analytic, B = 64: 12 608 image rows + 2 048 text rows,
12 layers, D = 384), and what that makes in TFLOP/s and TB/s against the summed kernel time.  Kernel times of concurrent streams overlap, so
these are rates PER LAUNCH while it shares the chip, not shares of the step.
usage: tools/in_situ_roofline.py <dir with *kernel_stats.csv> <profiled steps>"""
import csv, glob, sys
d = sys.argv[1]; steps = int(sys.argv[2])
f = glob.glob(d + '/**/*kernel_stats.csv', recursive=True) + glob.glob(d + '/*kernel_stats.csv')
rows = list(csv.DictReader(open(f[0])))
L, D, Hd, H = 12, 384, 1536, 6
Mi, Mt, Ni, Nt, B = 64 * 197, 64 * 32, 197, 32, 64
M = Mi + Mt
W = lambda o, i: 2 * o * i                      # bf16 weight bytes
# family -> (match substrings, flops per step, algorithmic bytes per step)
def gemm(rows_, n, k, extra_rows_bytes=0):       # A + C in bf16, plus extra per-row bytes (residual, second store, ...)
    return 2.0 * rows_ * n * k, 2.0 * rows_ * (k + n) + extra_rows_bytes
fam = {}
f1, b1 = gemm(M, 3 * D, D)
fam["NT bias (qkv fwd)"] = (["k_gemm_mfma<0, 0, unsigned short, 1>"], L * f1, L * (b1 + 2 * W(3 * D, D)))
fp, bp = gemm(M, D, D, 2.0 * M * D); f2, b2 = gemm(M, D, Hd, 2.0 * M * D)
fam["NT bias+residual (proj, fc2 fwd)"] = (["k_gemm_mfma<0, 0, unsigned short, 2>", "k_gemm_mfma<0, 0, unsigned short, 3>"], L * (fp + f2), L * (bp + b2 + 2 * (W(D, D) + W(D, Hd))))
fg, bg = gemm(M, Hd, D, 2.0 * M * Hd)
fam["NT GELU, 2 stores (fc1 fwd)"] = (["k_gemm_mfma<0, 0, unsigned short, 8>"], L * fg, L * (bg + 2 * W(Hd, D)))
fa, ba = gemm(M, D, Hd); fb, bb = gemm(M, D, D); fc, bc = gemm(M, D, 3 * D)
fam["NN plain (fc1, proj, qkv dX)"] = (["k_gemm_mfma<0, 1, unsigned short, 0>"], L * (fa + fb + fc), L * (ba + bb + bc + 2 * (W(Hd, D) + W(D, D) + W(3 * D, D))))
fm, bm = gemm(M, Hd, D, 2.0 * M * Hd)
fam["NN x gelu' (fc2 dX)"] = (["k_gemm_mfma<0, 1, unsigned short, 9>"], L * fm, L * (bm + 2 * W(D, Hd)))
wsum = 3 * D * D + D * D + 2 * Hd * D
fam["dW grouped + AdamW (all linears)"] = (["k_gemm_dw_spec", "k_gemm_dw_wide", "k_gemm_tn_grouped"], L * 2.0 * M * wsum,
                                         L * (2.0 * M * (3 * D + D + D + D + Hd + D + D + Hd) + 2 * wsum * (7 * 4 + 2)))
att = lambda b, n: (4.0 * b * H * n * n * 64, 2.0 * b * n * (3 * D + D))
fam["attention fwd"] = (["k_attn_fwd_mfma"], L * (att(B, Ni)[0] + att(B, Nt)[0]), L * (att(B, Ni)[1] + att(B, Nt)[1]))
fam["attention bwd"] = (["k_attn_bwd"], L * 2.5 * (att(B, Ni)[0] + att(B, Nt)[0]), L * 2.0 * M * (3 * D + D + D + 3 * D))
fam["LayerNorm fwd"] = (["k_ln_fwd_g"], 0.0, 2 * L * 2.0 * M * D * 2)
fam["LayerNorm bwd"] = (["k_ln_bwd_g"], 0.0, 2 * L * 2.0 * M * D * 4)
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"# {f[0]}: {steps} steps, kernel time {tot / 1e6 / steps:.3f} ms per step (summed over concurrent streams)")
print(f"{'family':36s} {'launches/step':>13s} {'avg us':>8s} {'ms/step':>8s} {'GFLOP/step':>11s} {'GB/step':>8s} {'TFLOP/s':>8s} {'TB/s':>6s} {'% of 2.5 PF':>11s} {'% of 8 TB/s':>11s}")
for name, (pats, flops, bytes_) in fam.items():
    sel = [r for r in rows if any(p in r['Name'] for p in pats)]
    if not sel:
        continue
    calls = sum(int(r['Calls']) for r in sel) / steps
    ns = sum(float(r['TotalDurationNs']) for r in sel) / steps
    tf, tb = flops / ns / 1e3, bytes_ / ns / 1e3
    print(f"{name:36s} {calls:13.0f} {ns / calls / 1e3:8.1f} {ns / 1e6:8.3f} {flops / 1e9:11.1f} {bytes_ / 1e9:8.2f} {tf:8.1f} {tb:6.2f} {100 * tf / 2500:10.1f}% {100 * tb / 8:10.1f}%")
