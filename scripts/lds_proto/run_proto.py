"""Driver of the LDS-staged prototype (scripts/lds_proto/proto.hip): builds the schedule on the
device with torch, runs the kernel through ctypes, checks the result bit for bit against the
product path (pygim_amd) and prints times + schedule statistics.

    python scripts/lds_proto/run_proto.py [--K 16] [--kc 512] [--nwc 14] [--nwl 2] [--shape reddit]
"""
import argparse
import ctypes
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from pygim_amd import synth  # noqa: E402


def plan_tiles(deg_sorted, n, KC, NWC, target, kmax=16):
    """Tiles over the length-sorted rows: (start position, K) with K in {1,2,4,8,16} chosen so that a
    consumer wave has about `target` steps per chunk."""
    tiles = []
    p = 0
    d = deg_sorted.tolist()
    while p < n:
        lam = d[p] * KC / float(n)
        est = lam + 1.3 * lam ** 0.5 + 0.6  # expected longest of 8 Poisson(lam) segments, roughly
        K = 1
        while K * 2 <= kmax and K * 2 * est <= target:
            K *= 2
        tiles.append((p, K))
        p += NWC * 8 * K
    return tiles


def build_schedule(rowptr, col, n, KC, NWC, target=44, kmax=16, window=False):
    dev = col.device
    rp = rowptr.to(torch.int64)
    deg = rp[1:] - rp[:-1]
    nnz = int(col.numel())
    order = torch.argsort(deg, descending=True, stable=True)  # sorted position -> row
    tiles = plan_tiles(deg[order].cpu(), n, KC, NWC, target, kmax)
    ntiles = len(tiles)
    t_start = torch.tensor([t[0] for t in tiles] + [n], device=dev)
    t_K = torch.tensor([t[1] for t in tiles], device=dev)
    pos = torch.empty(n, dtype=torch.int64, device=dev)
    pos[order] = torch.arange(n, device=dev)
    nchunks = (n + KC - 1) // KC
    # slot of a sorted position
    p = torch.arange(n, device=dev)
    tile = torch.bucketize(p, t_start, right=True) - 1
    q = p - t_start[tile]
    m = q // 8
    grp = q % 8
    wave = m % NWC
    j = m // NWC
    rm_base = torch.cumsum(t_K * NWC * 8, 0) - t_K * NWC * 8
    rowmap = torch.full((int((t_K * NWC * 8).sum()),), -1, dtype=torch.int64, device=dev)
    rowmap[rm_base[tile] + (wave * t_K[tile] + j) * 8 + grp] = order
    tileinfo = torch.stack([t_K, rm_base], 1).to(torch.int32).contiguous()
    del p, q, m
    # per entry
    row = torch.repeat_interleave(torch.arange(n, device=dev), deg)
    pe = pos[row]
    chunk = (col.to(torch.int64) // KC)
    seg = ((tile[pe] * nchunks + chunk) * NWC + wave[pe]) * 16 + j[pe]
    key = seg * 8 + grp[pe]
    tile_e = tile[pe]
    del pe
    # rank of the entry inside its (row, chunk) run (entries are sorted by row, then column)
    e = torch.arange(nnz, device=dev)
    rk = row * nchunks + chunk
    start = torch.ones(nnz, dtype=torch.bool, device=dev)
    start[1:] = rk[1:] != rk[:-1]
    first = torch.where(start, e, torch.zeros_like(e))
    first = torch.cummax(first, 0).values
    t = e - first
    del rk, start, first, e, row
    nseg = ntiles * nchunks * NWC * 16
    cnt = torch.bincount(key, minlength=nseg * 8).view(nseg, 8)
    if window:
        # window schedule: during phase j a lane group works on its row j or already on row j + 1
        c3 = cnt.view(-1, 16, 8)
        nw = c3.shape[0]
        F = torch.zeros((nw, 8), dtype=torch.int64, device=dev)
        Pm1 = torch.zeros(nw, dtype=torch.int64, device=dev)
        Pm2 = torch.zeros(nw, dtype=torch.int64, device=dev)
        S_all = torch.empty((nw, 16, 8), dtype=torch.int64, device=dev)
        P_all = torch.empty((nw, 16), dtype=torch.int64, device=dev)
        for jj in range(16):
            S = torch.maximum(F, Pm2[:, None])
            F = S + c3[:, jj, :]
            P = torch.maximum(Pm1, F.max(1).values)
            S_all[:, jj] = S
            P_all[:, jj] = P
            Pm2, Pm1 = Pm1, P
        del cnt, c3
        nph = P_all.clone()
        nph[:, 1:] -= P_all[:, :-1]
        assert int(nph.max()) < 256
        total_w = P_all[:, 15]
        step_base = torch.cumsum(total_w, 0) - total_w
        total_steps = int(total_w.sum())
        stream = torch.full(((total_steps + 512), 8), KC, dtype=torch.int16, device=dev).view(-1)
        wseg_e = seg // 16
        j_e = seg % 16
        g_e = key % 8
        t_abs = S_all.view(-1)[seg * 8 + g_e] + t
        prevP = torch.where(j_e > 0, P_all.view(-1)[wseg_e * 16 + torch.clamp(j_e - 1, min=0)], torch.zeros_like(t_abs))
        bit = (t_abs < prevP).to(torch.int64)
        idv = (col.to(torch.int64) - chunk * KC) + (bit << 15)
        stream[(step_base[wseg_e] + t_abs) * 8 + g_e] = (((idv + 32768) % 65536) - 32768).to(torch.int16)
        del idv, t_abs, prevP, bit, wseg_e, j_e, g_e, t, key, seg, chunk
        segn = nph.to(torch.uint8).contiguous()
        blk_off = torch.cat([step_base, torch.tensor([total_steps], device=dev)]).contiguous()
        nblk = nph  # per-tile statistics below count steps
        total_blk = total_steps
        unit_steps, unit_bytes = 1, 16
    else:
        unit_steps, unit_bytes = 2, 32
        nsteps = cnt.max(1).values
        del cnt
        nblk = (nsteps + 1) // 2
        assert int(nblk.max()) < 256
        blk_start = torch.cumsum(nblk, 0) - nblk
        total_blk = int(nblk.sum())
        # every block's first id carries its accumulator index j in bits 12..15 (flat kernel variant)
        stream = torch.empty(((total_blk + 256), 8, 2), dtype=torch.int16, device=dev)
        stream[:, :, 1] = KC
        stream[:, :, 0] = KC
        stream = stream.view(-1)
        spos = ((blk_start[seg] + t // 2) * 8 + (key % 8)) * 2 + (t % 2)
        stream[spos] = (col.to(torch.int64) - chunk * KC).to(torch.int16)
        del spos, t, key, seg, chunk
        segn = nblk.view(-1, 16).to(torch.uint8).contiguous()
        blk_off = torch.cat([blk_start.view(-1, 16)[:, 0], torch.tensor([total_blk], device=dev)]).contiguous()
    assert total_blk < 2 ** 32
    per_wg = blk_off[::NWC]
    per_wg = per_wg[1:] - per_wg[:-1]
    stats = {
        "ntiles": ntiles, "nchunks": nchunks, "total_steps": total_blk * unit_steps,
        "slot_eff": nnz / (total_blk * unit_steps * 8.0),
        "steps_per_tile": (nblk.view(ntiles, -1).sum(1) * unit_steps).tolist(),
        "nnz_per_tile": torch.bincount(tile_e, minlength=ntiles).tolist(),
        "K": [t[1] for t in tiles],
        "max_ids_bytes": int(per_wg.max()) * unit_bytes + 64,
    }
    return (blk_off.to(torch.uint32), segn, stream, rowmap.to(torch.int32), tileinfo, stats)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--target", type=int, default=44)
    ap.add_argument("--kmax", type=int, default=16)
    ap.add_argument("--flat", type=int, default=0)
    ap.add_argument("--clustered", type=int, default=0)
    ap.add_argument("--idb", type=int, default=14336)
    ap.add_argument("--kc", type=int, default=512)
    ap.add_argument("--nwc", type=int, default=14)
    ap.add_argument("--nwl", type=int, default=2)
    ap.add_argument("--shape", default="reddit")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--check", type=int, default=1)
    ap.add_argument("--tiles", default="", help="t0:count sub-range timing, comma separated")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = ctypes.CDLL(os.path.join(HERE, "libproto.so"))
    lib.proto_run.restype = ctypes.c_float
    n, nnz, dmax = synth.SHAPES[args.shape]
    rowptr, col = synth.make_csr(n, nnz, dmax, seed=0, device=dev, clustered=bool(args.clustered))
    h = 256
    X = synth.features(n, h, torch.float32, seed=0, device=dev)
    KC, NWC, NWL = args.kc, args.nwc, args.nwl
    t0 = time.time()
    blk_off, segn, stream, rowmap, tileinfo, st = build_schedule(rowptr, col, n, KC, NWC, args.target, args.kmax, window=bool(args.flat))
    torch.cuda.synchronize()
    print(f"schedule: {time.time() - t0:.1f}s tiles {st['ntiles']} chunks {st['nchunks']} max ids bytes {st['max_ids_bytes']} "
          f"steps {st['total_steps']} slot efficiency {st['slot_eff']:.3f} stream {stream.numel() * 2 / 1e6:.0f} MB")
    spt = st["steps_per_tile"]
    print("K per tile:", st["K"][:40], "...")
    assert st["max_ids_bytes"] <= args.idb, "id stream of one chunk exceeds its LDS buffer"
    print("steps per tile (first 8, median, last):", spt[:8], sorted(spt)[len(spt) // 2], spt[-1])
    npt = st["nnz_per_tile"]
    print("slot eff per tile (first 8, median, last):",
          [round(a / (8.0 * b), 3) for a, b in zip(npt[:8], spt[:8])],
          round(sorted(a / (8.0 * b) for a, b in zip(npt, spt))[len(spt) // 2], 3),
          round(npt[-1] / (8.0 * max(spt[-1], 1)), 3))
    nslices = h * 4 // 128
    nchunks = st["nchunks"]
    rows_pad = nchunks * KC
    Xs = torch.zeros((nslices, rows_pad, 32), dtype=torch.float32, device=dev)
    Xs[:, :n, :] = X.view(n, nslices, 32).permute(1, 0, 2)
    C = torch.zeros((n, h), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()

    def run(tile0, ntiles, iters):
        ms = lib.proto_run(args.flat, NWC, NWL, ctypes.c_void_p(Xs.data_ptr()), ctypes.c_int64(rows_pad * 128),
                           ctypes.c_void_p(blk_off.data_ptr()), ctypes.c_void_p(segn.data_ptr()),
                           ctypes.c_void_p(stream.data_ptr()), ctypes.c_void_p(rowmap.data_ptr()),
                           ctypes.c_void_p(tileinfo.data_ptr()),
                           ctypes.c_void_p(C.data_ptr()), ctypes.c_int64(h), nchunks, KC, args.idb, tile0, ntiles,
                           nslices, iters)
        return ms

    ms = run(0, st["ntiles"], args.iters)
    print(f"clustered {args.clustered} flat {args.flat} target {args.target} KC {KC} NWC {NWC} NWL {NWL}: all tiles {ms:.3f} ms  ({2 * nnz * h / ms / 1e9:.1f} TFLOP/s)")
    if args.check:
        ref = torch.zeros_like(C)
        # reference: torch CSR product in float64 on the device (integer-valued features: exact)
        A = torch.sparse_csr_tensor(rowptr.to(torch.int64), col.to(torch.int64),
                                    torch.ones(nnz, dtype=torch.float32, device=dev), size=(n, n))
        ref = torch.sparse.mm(A, X)
        bad = int((ref != C).sum())
        print("check vs torch.sparse.mm: mismatching elements", bad, "of", C.numel())
    for spec in [s for s in args.tiles.split(",") if s]:
        a, b = spec.split(":")
        a, b = int(a), int(b)
        ms = run(a, b, args.iters)
        steps = sum(spt[a:a + b])
        nz = sum(npt[a:a + b])
        wgs = b * nslices
        print(f"tiles {a}+{b}: {ms:.3f} ms, {wgs} WGs, steps {steps}, "
              f"{(ms * 1e-3) * 2.4e9 * min(256, wgs) / (steps * nslices):.1f} CU-cycles per step (2.4 GHz), "
              f"useful LDS rate {nz * 1024 / (ms * 1e-3) / 1e12:.1f} TB/s")


if __name__ == "__main__":
    main()
