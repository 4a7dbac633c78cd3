"""End-to-end inference stacks on the GPU: the aggregation inside GCN / SAGE / GIN
goes through the HIP backend (fused quantise -> SpMM -> dequantise) and must equal the same network with
the aggregation written in dense torch (int64 matmul of the quantised features)."""
import types

import numpy as np
import pytest
import torch

from conftest import random_csr
from pygim_amd import gnn, pim_ops, quantize
from pygim_amd.backend_pim import spmm as spmm_mod
from pygim_amd.sparse_tensor import SparseTensorShim

pytestmark = pytest.mark.gpu


def _free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class DenseAdj:
    """reference aggregator: same quantiser, exact integer product with a dense adjacency"""

    def __init__(self, dense_int64, dtype):
        self.a, self.dtype = dense_int64, dtype

    def mul(self, x_q):
        return (self.a.double() @ x_q.double()).to(x_q.dtype)  # exact: |sums| < 2^53


@pytest.mark.parametrize("name", ["gcn", "sage", "gin"])
def test_inference_stack_matches_dense_reference(rng, name):
    n, fin, h, ncls = 700, 48, 64, 7
    rowptr, col = random_csr(rng, n, n, 12, long_rows=[(3, 2000)])
    adj = SparseTensorShim(rowptr=torch.from_numpy(rowptr), col=torch.from_numpy(col), sparse_sizes=(n, n))
    dense = adj.to_dense(torch.float64).to(torch.int64).cuda()
    pim_ops.load("spmm")
    torch.ops.pim_ops.dpu_init_ranks(2)
    try:
        A = spmm_mod.prepare_pim_spmm(adj.to("cuda"), types.SimpleNamespace(data_type=torch.int32, sp_format="CSR", sp_parts=2,
                                                                               ds_parts=1, hidden_size=h))
        torch.manual_seed(3)
        model = {"gcn": gnn.GCN, "sage": gnn.SAGE, "gin": gnn.GIN}[name](fin, h, ncls, num_layers=3).cuda().eval()
        x = torch.randn(n, fin, device="cuda")
        with torch.no_grad():
            out = model(x, A, None)
            ref = model(x, DenseAdj(dense, torch.int32), None)
        assert out.shape == (n, ncls)
        assert torch.equal(out, ref)
    finally:
        torch.ops.pim_ops.dpu_release()


def test_row_split_adj_single_rank(rng):
    """the multi-GPU adjacency object with world_size 1 (the collective is the identity)"""
    from pygim_amd.dist import RowSplitAdj

    n, h = 500, 64
    rowptr, col = random_csr(rng, n, n, 10)
    adj = RowSplitAdj(torch.from_numpy(rowptr), torch.from_numpy(col), n, torch.int32, h)
    x = torch.randn(n, h, device="cuda")
    out, scale = adj.mul_quantized(x)
    s_ref, xq = quantize.symmetric_quantize(x, torch.int32)
    dense = SparseTensorShim(rowptr=torch.from_numpy(rowptr), col=torch.from_numpy(col), sparse_sizes=(n, n)).to_dense(torch.float64)
    prod = (dense.cuda() @ xq.double()).to(torch.int32)
    ref = prod * (1.0 * s_ref)
    assert torch.equal(out, ref) and torch.equal(scale, s_ref)
    assert torch.equal(adj.mul(xq), prod)
    adj._lib.release()


@pytest.mark.parametrize("tdt", [torch.int8, torch.int16, torch.int32, torch.float32])
def test_quantiser_steps_of_the_c_abi(rng, tdt):
    """pygim_quant_absmax / pygim_quantize / pygim_dequantize against the numpy restatement of models/quantize.py:20-42
    (oracle.symmetric_quantize): contiguous (float4 path) and strided / odd-sized inputs"""
    import oracle
    from pygim_amd import _lib
    from pygim_amd.pim_ops import DTYPE_CODE

    _lib.init_ranks(1)
    npdt = {torch.int8: np.int8, torch.int16: np.int16, torch.int32: np.int32, torch.float32: np.float32}[tdt]
    try:
        for rows, w, ld in ((300, 64, 64), (301, 7, 7), (50, 33, 40), (1, 4, 4), (0, 8, 8)):
            base = torch.from_numpy(rng.standard_normal((max(rows, 1), ld)).astype(np.float32) * 3).cuda()
            x = base[:rows, :w]
            bits = torch.zeros(1, dtype=torch.int32, device="cuda")
            _lib.quant_absmax(x.data_ptr(), ld, rows, w, bits.data_ptr())
            xq = torch.empty((rows, w), dtype=tdt, device="cuda")
            scale = torch.zeros((), dtype=torch.float32, device="cuda")
            _lib.quantize(DTYPE_CODE[tdt], x.data_ptr(), ld, rows, w, bits.data_ptr(), xq.data_ptr(), scale.data_ptr())
            out = torch.empty((rows, w), dtype=torch.float32, device="cuda")
            _lib.dequantize(DTYPE_CODE[tdt], xq.data_ptr(), rows * w, bits.data_ptr(), out.data_ptr())
            torch.cuda.synchronize()
            if rows == 0:
                assert int(bits.item()) == 0
                continue
            xs = x.cpu().numpy()
            s_ref, q_ref = oracle.symmetric_quantize(xs, npdt)
            assert bits.cpu().numpy().view(np.float32)[0] == np.float32(np.abs(xs).max())
            assert np.float32(scale.item()) == s_ref and np.array_equal(xq.cpu().numpy(), q_ref), (rows, w, ld)
            assert np.array_equal(out.cpu().numpy(), oracle.symmetric_dequantize(q_ref, 1.0, s_ref))
    finally:
        _lib.release()


@pytest.mark.parametrize("tdt", [torch.int8, torch.int32, torch.float32])
def test_row_shard_adj_single_rank_equals_fused_call(rng, tdt):
    """RowShardAdj with world_size 1 (collectives are identities; columns re-based to the padded layout): the stepwise
    quantise -> exchange -> aggregate -> dequantise equals the one-call fused path bit for bit"""
    from pygim_amd.dist import RowShardAdj, RowSplitAdj

    n, h = 600, 64
    rowptr, col = random_csr(rng, n, n, 10, long_rows=[(9, 1500)])
    x = torch.randn(n, h, device="cuda")
    sh = RowShardAdj(torch.from_numpy(rowptr), torch.from_numpy(col), n, tdt, h)
    out, bits = sh.mul_quantized(x)
    ref_adj = RowSplitAdj(torch.from_numpy(rowptr), torch.from_numpy(col), n, tdt, h)
    ref, scale = ref_adj.mul_quantized(x)
    assert torch.equal(out, ref)
    assert bits.cpu().numpy().view(np.float32)[0] == np.float32(x.abs().max().item())
    sh.engine._lib.release()


def _run_inference(world, extra, env_extra=None):
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **(env_extra or {}))
    args = ["--dataset=PubMed", "--model=gcn", "--num_layers=3", "--hidden_size=64", "--repeat=2"] + extra
    if world == 1:
        cmd = [sys.executable, os.path.join(root, "inference.py")] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr=127.0.0.1",
               f"--master-port={_free_port()}", os.path.join(root, "inference.py")] + args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    sums = [float(ln.split(":")[1]) for ln in r.stdout.splitlines() if ln.startswith("[DATA]logits_checksum")]
    accs = [float(ln.split(":")[1]) for ln in r.stdout.splitlines() if ln.startswith("Test_acc")]
    assert len(sums) == 2 and len(accs) == 2, r.stdout[-2000:]
    return sums, accs


def test_inference_gpus_flag_starts_its_own_ranks():
    """`python inference.py --gpus 2 ...` as a plain command (round 5, as bench.py --gpus N: the ranks are child processes of
    torch.distributed.run started by pygim_amd/launch.py) gives what the same command under an explicit launcher gives"""
    plain, _ = _run_inference(1, ["--data_type=INT32", "--gpus=2"], {"PYGIM_BENCH_BACKEND": "gloo"})
    launched, _ = _run_inference(2, ["--data_type=INT32"], {"PYGIM_BENCH_BACKEND": "gloo"})
    assert plain == launched, (plain, launched)


@pytest.mark.parametrize("dtype", ["INT8", "INT32"])
def test_row_sharded_inference_two_ranks_equals_one_rank(dtype):
    """inference.py under torch.distributed.run with 2 ranks (gloo, both on this one GPU: a logic check of the N > 1 path --
    row-sharded activations, MAX all-reduce of the scale, all-gather of the quantised blocks) gives the one-rank logits"""
    one, acc1 = _run_inference(1, [f"--data_type={dtype}"])
    two, acc2 = _run_inference(2, [f"--data_type={dtype}"], {"PYGIM_BENCH_BACKEND": "gloo"})
    # the aggregation is bit-identical per row; the dense layers run on [rows_r, h] blocks instead of [N, h], where the
    # GEMM library may tile (and so round) differently: 1e-5 relative, the floating-point bar of the path
    assert abs(one[0] - two[0]) <= 1e-5 * abs(one[0]) and abs(one[1] - two[1]) <= 1e-5 * abs(one[1]), (one, two)
    assert abs(acc1[0] - acc2[0]) < 2e-3


@pytest.mark.parametrize("name", ["INT8", "INT16", "INT32", "FLT32"])
def test_quantiser_golden_vectors_on_gpu(name):
    """tests/golden/quant_gcn_*.npz through the C ABI: the fused call and the three steps"""
    import os

    from pygim_amd import _lib
    from pygim_amd.pim_ops import DTYPE_CODE

    z = np.load(os.path.join(os.path.dirname(__file__), "golden", f"quant_gcn_{name}.npz"))
    tdt = {"INT8": torch.int8, "INT16": torch.int16, "INT32": torch.int32, "FLT32": torch.float32}[name]
    n, h = z["x"].shape
    d = lambda a: torch.from_numpy(a).cuda()
    rp, cl, x = d(z["rowptr"].astype(np.int32)), d(z["col"].astype(np.int32)), d(z["x"])
    _lib.init_ranks(1)
    try:
        hd = _lib.group_create(_lib.CSR, DTYPE_CODE[tdt], [rp.data_ptr()], [cl.data_ptr()], None, [n], [n], [cl.numel()], [1], [h], h)
        out = torch.empty((n, h), dtype=torch.float32, device="cuda")
        scale = torch.zeros((), dtype=torch.float32, device="cuda")
        _lib.quant_spmm_run(hd, x.data_ptr(), h, out.data_ptr(), scale.data_ptr())
        torch.cuda.synchronize()
        import oracle

        def same(got, want, unit):
            # integers: bit-exact.  FLT32 keeps the quantised values as floats: sums of ~2^19-sized integers round, and the
            # 1200-entry row is summed by a whole wave in another order -> the 1e-5 bound relative to |A| . |x_q|
            if name != "FLT32":
                return np.array_equal(got, want)
            bound = 1e-5 * oracle.spmm_csr(z["rowptr"], z["col"], None, np.abs(z["xq"])).astype(np.float64) * unit
            return bool(np.all(np.abs(got.astype(np.float64) - want.astype(np.float64)) <= bound))

        assert np.float32(scale.item()) == z["scale"] and same(out.cpu().numpy(), z["out"], float(z["scale"]))
        bits = torch.zeros(1, dtype=torch.int32, device="cuda")
        xq = torch.empty((n, h), dtype=tdt, device="cuda")
        _lib.quant_absmax(x.data_ptr(), h, n, h, bits.data_ptr())
        _lib.quantize(DTYPE_CODE[tdt], x.data_ptr(), h, n, h, bits.data_ptr(), xq.data_ptr())
        oq = torch.empty((n, h), dtype=tdt, device="cuda")
        _lib.spmm_run_group(hd, [xq.data_ptr()], oq.data_ptr())
        out2 = torch.empty((n, h), dtype=torch.float32, device="cuda")
        _lib.dequantize(DTYPE_CODE[tdt], oq.data_ptr(), n * h, bits.data_ptr(), out2.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(xq.cpu().numpy(), z["xq"]) and same(oq.cpu().numpy(), z["out_q"], 1.0)
        assert same(out2.cpu().numpy(), z["out"], float(z["scale"]))
        _lib.group_free(hd)
    finally:
        _lib.release()


@pytest.mark.parametrize("world", [2, 4])
def test_multi_gpu_layer_with_hip_engines_over_gloo(world):
    """pygim_amd/dist.py end to end with the real engines (C ABI, HIP kernels), `world` ranks over gloo on this one GPU:
    row / column / feature splits, every 2-D grid, row-sharded quantised aggregation -- all equal to the oracle"""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr=127.0.0.1",
           f"--master-port={_free_port()}", os.path.join(root, "tests", "dist_driver.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert all(f"OK rank {k}" in r.stdout for k in range(world)), r.stdout[-2000:]


class _Capture:
    """adjacency wrapper that keeps what every aggregation saw and returned"""

    def __init__(self, adj):
        self.adj, self.dtype, self.calls = adj, adj.dtype, []

    def mul_quantized(self, x):
        out, scale = self.adj.mul_quantized(x)
        self.calls.append((x.detach().clone(), out.detach().clone()))
        return out, scale


def test_reddit_gcn_three_layers_h256_flt32_full_size():
    """configs[3] on one rank at full size: Reddit-shaped graph (232 965 nodes, 114.6 M edges), 3-layer GCN, h = 256, FLT32
    adjacency.  Every layer's aggregation -- quantise (values kept as floats) -> A . x_q -> dequantise, one fused device
    call -- is compared on sampled rows (first, last, longest) with the oracle's quantiser and CSR loop, within the
    floating-point bar of the path (1e-5 of |A| . |x_q| . scale: sums of ~500 terms of size 2^19 round in float)."""
    import oracle
    from pygim_amd import synth
    from pygim_amd.dist import RowSplitAdj

    dev = torch.device("cuda", 0)
    n, nnz, dmax = synth.SHAPES["reddit"]
    h, fin, ncls = 256, 128, 41
    rowptr, col = synth.make_csr(n, nnz, dmax, seed=0, device=dev)
    adj = _Capture(RowSplitAdj(rowptr, col, n, torch.float32, h))
    try:
        torch.manual_seed(1)
        model = gnn.GCN(fin, h, ncls, num_layers=3).to(dev).eval()
        torch.manual_seed(0)
        x = torch.randn(n, fin, device=dev)
        with torch.no_grad():
            logits = model(x, adj, None)
            again = model(x, _Capture(adj.adj), None)
        assert logits.shape == (n, ncls) and torch.isfinite(logits).all()
        assert torch.equal(logits, again)  # deterministic end to end
        assert len(adj.calls) == 3
        deg = (rowptr[1:] - rowptr[:-1])
        longest = int(torch.argmax(deg))
        rp = rowptr.to(torch.int64)
        for layer, (xin, out) in enumerate(adj.calls):
            s_ref, xq = oracle.symmetric_quantize(xin.cpu().numpy(), np.float32)
            for r0, r1 in ((0, 64), (n - 64, n), (longest, longest + 1)):
                lo, hi = int(rp[r0]), int(rp[r1])
                sub_rp = (rp[r0:r1 + 1] - lo).cpu().numpy().astype(np.int32)
                sub_col = col[lo:hi].cpu().numpy()
                want = oracle.symmetric_dequantize(oracle.spmm_csr(sub_rp, sub_col, None, xq), 1.0, s_ref)
                bound = 1e-5 * oracle.spmm_csr(sub_rp, sub_col, None, np.abs(xq)).astype(np.float64) * float(s_ref)
                got = out[r0:r1].cpu().numpy().astype(np.float64)
                assert np.all(np.abs(got - want.astype(np.float64)) <= bound + 1e-30), (layer, r0)
    finally:
        adj.adj._lib.release()


def test_reddit_gcn_h256_flt32_two_ranks_equal_one_rank():
    """configs[3], the sp_parts row split: inference.py on the Reddit-shaped graph, GCN 3 layers, h = 256, FLT32, with 2 ranks
    (gloo, both on this GPU: row-sharded activations, MAX all-reduce of the scale, all-gather of the quantised blocks)
    gives the one-rank logits"""
    common = ["--dataset=Reddit", "--hidden_size=256", "--data_type=FLT32", "--version=spmm",
              "--lib_path=./backend_pim/spmm_default/build/libbackend_pim.so"]
    one, _ = _run_inference(1, common)
    two, _ = _run_inference(2, common, {"PYGIM_BENCH_BACKEND": "gloo"})
    assert abs(one[0] - two[0]) <= 1e-5 * abs(one[0]) and abs(one[1] - two[1]) <= 1e-5 * abs(one[1]), (one, two)


@pytest.mark.parametrize("tdt", [torch.int8, torch.int16, torch.int32, torch.float32])
@pytest.mark.parametrize("h", [256, 100, 37])
def test_fused_quantised_aggregation_equals_the_three_steps(rng, tdt, h):
    """pygim_quant_spmm_run on a graph with several column panels, wave-cooperative (long) items and empty rows: the fused
    path (absmax -> quantising slice-major pack -> sweep whose last item per row dequantises) against the unfused three
    steps (panel_mode = 2 at run time keeps the group's plan but sends the call down quantise / product / dequantise) --
    bit for bit for the integer types; FLT32 sums round, so there the bar is 1e-5 of |A| . |x_q| . scale"""
    from pygim_amd import _lib
    from pygim_amd.pim_ops import DTYPE_CODE

    n = 9000
    rowptr, col = random_csr(rng, n, n, 30, long_rows=[(5, 6000), (77, 900), (n - 1, 2500)], empty_frac=0.05)
    _lib.init_ranks(1)
    old = _lib.set_tunable("panel_bytes", 128 * 1024)  # 1024 columns per panel -> 9 panels on this small graph
    force = _lib.set_tunable("panel_mode", 1)
    try:
        rp, cl = torch.from_numpy(rowptr).cuda(), torch.from_numpy(col).cuda()
        hd = _lib.group_create(_lib.CSR, DTYPE_CODE[tdt], [rp.data_ptr()], [cl.data_ptr()], None, [n], [n], [cl.numel()], [1], [h], h)
        plan = _lib.group_plan(hd)
        assert plan["n_panels"] >= 8 and plan["n_coop_items"] > 0
        x = torch.from_numpy(rng.standard_normal((n, h + 3)).astype(np.float32) * 2.5).cuda()[:, :h]  # strided rows
        out = torch.full((n, h), float("nan"), dtype=torch.float32, device="cuda")
        scale = torch.zeros((), dtype=torch.float32, device="cuda")
        _lib.quant_spmm_run(hd, x.data_ptr(), x.stride(0), out.data_ptr(), scale.data_ptr())
        torch.cuda.synchronize()
        _lib.set_tunable("panel_mode", 2)  # unfused: three steps
        ref = torch.full((n, h), float("nan"), dtype=torch.float32, device="cuda")
        scale2 = torch.zeros((), dtype=torch.float32, device="cuda")
        _lib.quant_spmm_run(hd, x.data_ptr(), x.stride(0), ref.data_ptr(), scale2.data_ptr())
        torch.cuda.synchronize()
        _lib.set_tunable("panel_mode", 1)
        assert torch.equal(scale, scale2) and not torch.isnan(out).any()
        if tdt != torch.float32:
            assert torch.equal(out, ref)
        else:
            import oracle

            s_ref, xq = oracle.symmetric_quantize(x.cpu().numpy(), np.float32)
            bound = 1e-5 * oracle.spmm_csr(rowptr, col, None, np.abs(xq)).astype(np.float64) * float(s_ref)
            assert np.all(np.abs(out.cpu().numpy().astype(np.float64) - ref.cpu().numpy().astype(np.float64)) <= 2 * bound + 1e-30)
        # the two-step form used between ranks: quantise (exchange) -> product with the dequantising store
        bits = torch.zeros(1, dtype=torch.int32, device="cuda")
        _lib.quant_absmax(x.data_ptr(), x.stride(0), n, h, bits.data_ptr())
        xq = torch.empty((n, h), dtype=tdt, device="cuda")
        _lib.quantize(DTYPE_CODE[tdt], x.data_ptr(), x.stride(0), n, h, bits.data_ptr(), xq.data_ptr())
        out3 = torch.full((n, h), float("nan"), dtype=torch.float32, device="cuda")
        _lib.spmm_run_dequant(hd, xq.data_ptr(), h, out3.data_ptr(), bits.data_ptr())
        torch.cuda.synchronize()
        if tdt != torch.float32:
            assert torch.equal(out3, ref)
        elif h % 4 == 0:
            assert torch.equal(out3, out)  # same quantiser, same sweep order as the one-call form
        else:
            assert np.all(np.abs(out3.cpu().numpy().astype(np.float64) - ref.cpu().numpy().astype(np.float64)) <= 2 * bound + 1e-30)
        _lib.group_free(hd)
    finally:
        _lib.set_tunable("panel_bytes", old)
        _lib.set_tunable("panel_mode", force)
        _lib.release()


@pytest.mark.parametrize("tdt", [torch.int8, torch.float32])
def test_gcn_with_the_epilogue_in_the_last_store(rng, tdt):
    """fuse_post: a GCN layer's + bias -> BatchNorm (eval) -> ReLU as the per-column epilogue of the aggregation's fused
    store.  Same mathematics as the three torch passes, another rounding sequence: 1e-5 relative on the logits; the
    epilogue itself (pygim_quant_spmm_run_post) against the plain call + torch arithmetic in the same order: bit for bit."""
    from pygim_amd.dist import RowSplitAdj

    n, fin, h, ncls = 2500, 32, 96, 7
    rowptr, col = random_csr(rng, n, n, 14, long_rows=[(3, 1800)], empty_frac=0.05)
    adj = RowSplitAdj(torch.from_numpy(rowptr), torch.from_numpy(col), n, tdt, h)
    try:
        torch.manual_seed(5)
        model = gnn.GCN(fin, h, ncls, num_layers=3).cuda().eval()
        for bn in model.bns:  # non-trivial statistics
            bn.running_mean.uniform_(-0.5, 0.5)
            bn.running_var.uniform_(0.5, 2.0)
            bn.weight.data.uniform_(0.5, 1.5)
            bn.bias.data.uniform_(-0.3, 0.3)
        for conv in model.convs:
            conv.bias.data.uniform_(-0.2, 0.2)
        x = torch.randn(n, fin, device="cuda")
        with torch.no_grad():
            ref = model(x, adj, None)
            model.fuse_post = True
            out = model(x, adj, None)
            # (FLT32 quantises to 2^20 levels: a last-bit difference before the next layer's round() moves values by one level,
            #  2e-6 of max|x|, and three layers of 14-term sums carry that to ~1e-4; INT8's 32 levels never flip here)
            tol = 1e-5 if tdt == torch.int8 else 1e-3
            assert torch.allclose(out, ref, rtol=tol, atol=tol), float((out - ref).abs().max())
            # the epilogue alone, bit for bit
            y = torch.randn(n, h, device="cuda")
            a, b = torch.rand(h, device="cuda") + 0.5, torch.randn(h, device="cuda")
            plain, _ = adj.mul_quantized(y)
            fused, _ = adj.mul_quantized(y, post=(a, b, True))
            assert torch.equal(fused, torch.relu(a * plain + b))
            fused2, _ = adj.mul_quantized(y, post=(a, b, False))
            assert torch.equal(fused2, a * plain + b)
    finally:
        adj._lib.release()


def test_quantised_entry_points_reject_bad_arguments(rng):
    """error behaviour of the round-2 entry points: status code + message, nothing launched"""
    from pygim_amd import _lib

    n, h = 64, 16
    rowptr, col = random_csr(rng, n, n, 4)
    rp, cl = torch.from_numpy(rowptr).cuda(), torch.from_numpy(col).cuda()
    _lib.init_ranks(1)
    try:
        hd = _lib.group_create(_lib.CSR, _lib.INT8, [rp.data_ptr()], [cl.data_ptr()], None, [n], [n], [cl.numel()], [1], [h], h)
        x = torch.randn(n, h, device="cuda")
        out = torch.empty(n, h, device="cuda")
        a = torch.ones(h, device="cuda")
        bits = torch.zeros(1, dtype=torch.int32, device="cuda")
        xq = torch.zeros((n, h), dtype=torch.int8, device="cuda")
        for call, what in (
                (lambda: _lib.quant_spmm_run(hd, x.data_ptr(), h, out.data_ptr(), 0, 0, a.data_ptr(), 0, True), "col_mul and col_add come together"),
                (lambda: _lib.quant_spmm_run(hd, x.data_ptr(), h - 1, out.data_ptr()), "bad X / out / ldx"),
                (lambda: _lib.quant_spmm_run(hd, x.cpu().data_ptr(), h, out.data_ptr()), "device pointers"),
                (lambda: _lib.spmm_run_dequant(hd, xq.data_ptr(), h - 1, out.data_ptr(), bits.data_ptr()), "bad Xq"),
                (lambda: _lib.spmm_run_dequant(hd, xq.cpu().data_ptr(), h, out.data_ptr(), bits.data_ptr()), "device pointers"),
                (lambda: _lib.spmm_run_dequant(12345, xq.data_ptr(), h, out.data_ptr(), bits.data_ptr()), "unknown group handle"),
                (lambda: _lib.group_plan(12345), "unknown group handle"),
                (lambda: _lib.group_kernel_events(12345, True), "unknown group handle")):
            with pytest.raises(_lib.PygimError) as e:
                call()
            assert what in str(e.value), (what, str(e.value))
        _lib.group_free(hd)
        hd64 = _lib.group_create(_lib.CSR, _lib.INT64, [rp.data_ptr()], [cl.data_ptr()], None, [n], [n], [cl.numel()], [1], [h], h)
        with pytest.raises(_lib.PygimError) as e:
            _lib.spmm_run_dequant(hd64, xq.data_ptr(), h, out.data_ptr(), bits.data_ptr())
        assert "INT8/INT16/INT32/FLT32" in str(e.value)
        _lib.group_free(hd64)
    finally:
        _lib.release()


def test_forward_pass_replays_from_a_hip_graph():
    """inference.py --graph 1: the whole forward pass (torch dense layers + the fused aggregation calls of the C ABI) captured
    into a HIP graph after two warm-up runs and replayed; same logits checksum as the eager run"""
    eager, _ = _run_inference(1, ["--data_type=INT8", "--graph=0"])
    graph, _ = _run_inference(1, ["--data_type=INT8", "--graph=1"])
    assert eager == graph, (eager, graph)
