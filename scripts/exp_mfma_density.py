#!/usr/bin/env python3
"""The MFMA clause of north_star, measured: "MFMA only where the LDS-resident feature tile is a true dense panel
contraction".  For the bench's two synthetic graphs -- and any processed/*.pt graph found under --datadir -- how dense are the blocks a
matrix-core contraction would have to multiply, and what would it cost at the dense MFMA peaks against what the code-stream kernel takes?

A workgroup tile is R = 1 821 rows x a 128-column chunk of X (the LDS-resident panel).  A matrix-core product of that block multiplies ALL
R x 128 cells (v_mfma_f32_32x32x2_f32 for exact FLT32 sums in another order; v_mfma_i32_32x32x32_i8 for INT8, exact), whatever is stored
in them, so its cost is the DENSE flop count of the blocks it is applied to.  Reported per graph:
  * the distribution of stored entries over (tile, chunk) blocks by block density;
  * at MFMA granularity (32 x 32 cells): non-empty blocks and entries per non-empty block;
  * time of the dense contraction over every TOUCHED (tile, chunk) block at the dense peaks (MI355X_MICROARCH.md: fp32 matrix 157.3 TFLOP/s,
    int8 ~5 POP/s dense) and over only the blocks above a density threshold (the rest left to the code stream), against the measured kernel.
exp_mfma_density.py [--datadir DIR]"""
import argparse, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pygim_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--datadir", default=os.path.join(ROOT, "data"))
ap.add_argument("--h", type=int, default=256)
args = ap.parse_args()
dev = torch.device("cuda", 0) if torch.cuda.is_available() else torch.device("cpu")
R, KC, FP32_PEAK, I8_PEAK = 1821, 128, 157.3e12, 5.0e15
MEASURED = {"uniform": 2.07e-3, "clustered": 1.06e-3}   # k_lds_code8_f32, profiles/r04_exp_code_geo.txt


def analyse(name, rowptr, col, n, measured):
    nnz = col.numel()
    deg = (rowptr[1:] - rowptr[:-1]).long()
    row = torch.repeat_interleave(torch.arange(n, device=col.device), deg)
    nch = -(-n // KC)
    key = (row // R) * nch + col.long() // KC
    blocks, cnt = torch.unique(key, return_counts=True)
    tiles = -(-n // R)
    rows_of = torch.full((tiles,), R, device=col.device)
    rows_of[-1] = n - R * (tiles - 1)
    dens = cnt.double() / (rows_of[blocks // nch].double() * KC)
    print(f"== {name}: N {n}, nnz {nnz}, {tiles} tiles of {R} rows x {nch} chunks of {KC} columns; {blocks.numel()} touched (tile, chunk) blocks "
          f"= {100.0 * blocks.numel() / (tiles * nch):.1f} % of all; overall density {nnz / n / n:.5f}")
    print("   block density      blocks   share of entries")
    edges = [0, 0.005, 0.01, 0.02, 0.05, 0.1, 0.2, 0.5, 1.01]
    for lo, hi in zip(edges[:-1], edges[1:]):
        m = (dens >= lo) & (dens < hi)
        print(f"   [{lo:5.3f}, {hi:5.3f})  {int(m.sum()):9d}   {100.0 * float(cnt[m].sum()) / nnz:6.2f} %")
    # MFMA granularity: 32 x 32 cells
    k32 = (row // 32) * (-(-n // 32)) + col.long() // 32
    b32, c32 = torch.unique(k32, return_counts=True)
    print(f"   32 x 32 cells: {b32.numel()} non-empty blocks, {nnz / b32.numel():.2f} stored entries per non-empty block (of 1 024 cells)")
    dense_flops_all = 2.0 * float((rows_of[blocks // nch].double() * KC).sum()) * args.h
    useful = 2.0 * nnz * args.h
    print(f"   dense contraction of every touched block: {dense_flops_all / 1e12:8.2f} TFLOP ({dense_flops_all / useful:6.1f} x the useful flops) "
          f"-> {dense_flops_all / FP32_PEAK * 1e3:8.2f} ms at the fp32 matrix peak, {dense_flops_all / I8_PEAK * 1e3:7.3f} ms at the int8 peak; "
          f"code stream measured {measured * 1e3:.2f} ms (f32)")
    for thr in (0.05, 0.1, 0.2):
        m = dens >= thr
        fl = 2.0 * float((rows_of[blocks[m] // nch].double() * KC).sum()) * args.h
        share = float(cnt[m].sum()) / nnz
        print(f"   only blocks of density >= {thr:4.2f}: {100 * share:6.2f} % of the entries, {fl / FP32_PEAK * 1e3:8.3f} ms fp32 MFMA / {fl / I8_PEAK * 1e3:7.4f} ms int8 MFMA "
              f"for them (the code stream spends ~{share * measured * 1e3:.3f} ms on the same entries)")


n, nnz, dmax = synth.SHAPES["reddit"]
for clustered in (False, True):
    rp, col = synth.make_csr(n, nnz, dmax, seed=0, device=dev, clustered=clustered)
    analyse("reddit-shaped synthetic, " + ("clustered (columns ~ N(row, N/100))" if clustered else "uniform columns"), rp, col, n,
            MEASURED["clustered" if clustered else "uniform"])
    del rp, col
found = sorted(glob.glob(os.path.join(args.datadir, "**", "processed", "*.pt"), recursive=True))
if not found:
    print(f"== no processed/*.pt under {args.datadir}: no real dataset on this box (no network); the loader for them is pygim_amd/datasets.py load_processed")
for path in found:
    from pygim_amd import datasets
    try:
        rowptr, col, nn = datasets.load_processed(path)
        analyse(path, torch.from_numpy(rowptr).to(dev), torch.from_numpy(col).to(dev), nn, float("nan"))
    except Exception as e:  # noqa: BLE001
        print(f"== {path}: {e}")
