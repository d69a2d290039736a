#!/usr/bin/env python3
"""rocprofv3 PMC counters per kernel for a workload of this repo, one pass per counter set (counters in their own runs with
--kernel-trace only: see the gpurun rules), summarised as text + JSON.

    python tools/pmc_kernels.py topk   [out_prefix]     # score_topk: Baby full evaluation + a 65,536 x 500,000 block
    python tools/pmc_kernels.py linear [out_prefix]     # 4096 -> 64 projection fwd / dW / dX at Baby, Clothing, C5 item counts
    python tools/pmc_kernels.py spmm   [out_prefix]     # CSR SpMM on the config-5 graphs (full / 80 %-pruned / item-item), d = 64 and the
                                                        # feature slices d = 32 / 8: memory-side counters (requests by size, TLB, TA / TCP stalls)

Per kernel (matched by substring, grouped by grid size): calls, mean duration, mean counter values, and the derived
MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CU_CYCLES-equivalent: duration x clock x SIMDs), reported both ways."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SETS = [
    ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY"],
    ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT"],
    ["SQ_INSTS_MFMA", "SQ_BUSY_CU_CYCLES", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU", "GRBM_GUI_ACTIVE"],
    ["TCC_HIT_sum", "TCC_MISS_sum"],
]


def child_topk():
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    from mmrec_amd import hip_ops, synth
    dev = torch.device("cuda:0")
    nu, ni, eu, ei = synth.shaped_edges("baby", seed=0)
    r, c, v = synth.sym_norm_coo(eu, ei, nu, ni)
    g = hip_ops.CsrGraph.from_coo_host(np.stack([r, c]), v, nu + ni, nu + ni, dev, symmetric=True)
    E0 = torch.empty(nu + ni, 64, device=dev)
    torch.nn.init.xavier_uniform_(E0[:nu]), torch.nn.init.xavier_uniform_(E0[nu:])
    E = hip_ops.lightgcn_mean(g, E0, 2)
    rp, col = hip_ops.mask_to_csr(np.stack([eu, ei]), nu, dev)
    U, I = E[:nu].contiguous(), E[nu:].contiguous()
    gen = torch.Generator(device=dev).manual_seed(9)
    nq, nc = 65536, 500000
    common = torch.randn(64, device=dev, generator=gen) * 0.05
    Q = torch.randn(nq, 64, device=dev, generator=gen) * 0.03 + common
    C = torch.randn(nc, 64, device=dev, generator=gen) * 0.03 + common
    key = np.unique(np.repeat(np.arange(nq), 8).astype(np.int64) * nc + np.random.default_rng(2).integers(0, nc, nq * 8))
    rp2, col2 = hip_ops.mask_to_csr(np.stack([key // nc, key % nc]), nq, dev)
    for _ in range(3):
        hip_ops.score_topk(U, I, 50, rp, col)
        hip_ops.score_topk(Q, C, 50, rp2, col2)
    torch.cuda.synchronize()


SPMM_SETS = [
    ["FETCH_SIZE"], ["WRITE_SIZE"], ["TCC_HIT_sum", "TCC_MISS_sum"],
    ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum"], ["TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"],
    ["TCC_EA0_RDREQ_DRAM_sum", "TCC_REQ_sum"], ["TCC_EA0_RDREQ_LEVEL_sum", "TCC_TAG_STALL_sum"],
    ["TCP_UTCL1_TRANSLATION_MISS_sum", "TCP_UTCL1_TRANSLATION_HIT_sum"],
    ["TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum"],
    ["TCP_TCC_READ_REQ_LATENCY_sum", "TCP_PENDING_STALL_CYCLES_sum"],
    ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VMEM_RD",
     "SQ_INST_LEVEL_VMEM"],
    ["TA_TA_BUSY_sum", "TA_ADDR_STALLED_BY_TC_CYCLES_sum"], ["TA_DATA_STALLED_BY_TC_CYCLES_sum", "TA_BUSY_avr"],
]


def graphs(dev):
    """the three SpMM operands of a config-5 step: the full user-item graph, its 80 %-pruned training graph, an item-item kNN-like graph"""
    import numpy as np
    from mmrec_amd import hip_ops, synth
    out = {}
    nu, ni, eu, ei = synth.shaped_edges("c5", seed=0)
    n = nu + ni
    r, c, v = synth.sym_norm_coo(eu, ei, nu, ni)
    out["c5_full_20M"] = (hip_ops.CsrGraph.from_coo_host(np.stack([r, c]), v, n, n, dev, symmetric=True), n)
    rng = np.random.default_rng(0)
    keep = np.sort(rng.choice(eu.shape[0], eu.shape[0] // 5, replace=False))
    r2, c2, v2 = synth.sym_norm_coo(eu[keep], ei[keep], nu, ni)
    out["c5_pruned_4M"] = (hip_ops.CsrGraph.from_coo_host(np.stack([r2, c2]), v2, n, n, dev, symmetric=True), n)
    rows = np.repeat(np.arange(ni), 20)
    cols = rng.integers(0, ni, rows.shape[0])
    out["c5_item_item_10M"] = (hip_ops.CsrGraph.from_coo_host(np.stack([rows, cols]), np.full(rows.shape[0], 0.05, np.float32),
                                                              ni, ni, dev), ni)
    return out


def child_spmm():
    import torch
    sys.path.insert(0, ROOT)
    from mmrec_amd import hip_ops
    dev = torch.device("cuda:0")
    for name, (g, n_x) in graphs(dev).items():
        for d in (64, 32, 8):
            x = torch.rand(n_x, d, device=dev) - 0.5
            y = torch.empty(g.n_rows, d, device=dev)
            for _ in range(3):
                hip_ops.spmm_raw(g, x, Y=y)
            torch.cuda.synchronize()
        print("[pmc-spmm] %s: rows %d nnz %d chunks %d" % (name, g.n_rows, g.nnz, g.n_chunks), file=sys.stderr, flush=True)


def child_linear():
    import torch
    sys.path.insert(0, ROOT)
    from mmrec_amd import hip_ops
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(0)
    for n in (7050, 23033, 500000):
        X = torch.rand(n, 4096, device=dev, generator=gen).requires_grad_()
        W = (torch.rand(64, 4096, device=dev, generator=gen) - 0.5).requires_grad_()
        b = torch.zeros(64, device=dev, requires_grad=True)
        G = torch.rand(n, 64, device=dev, generator=gen) - 0.5
        for _ in range(3):
            X.grad = W.grad = b.grad = None
            hip_ops.linear(X, W, b).backward(G)
        torch.cuda.synchronize()
        del X, W, b, G
        torch.cuda.empty_cache()


def main():
    what = sys.argv[1]
    if what.startswith("child_"):
        return {"child_topk": child_topk, "child_linear": child_linear, "child_spmm": child_spmm}[what]()
    prefix = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "pmc_" + what)
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    tmp = tempfile.mkdtemp(prefix="mmrec_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    agg = {}   # (kernel, grid) -> {counter: [values]}, plus durations
    for si, counters in enumerate(SPMM_SETS if what == "spmm" else SETS):
        d = os.path.join(tmp, "set%d" % si)
        cmd = [exe, "--kernel-trace", "--pmc"] + counters + ["--output-format", "csv", "-d", d, "-o", "pm", "--",
                                                             sys.executable, os.path.abspath(__file__), "child_" + what]
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=900)
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not files:
            print("set %d failed (rc %d): %s" % (si, r.returncode, r.stderr[-400:]), flush=True)
            continue
        per_dispatch = {}   # a dispatch appears once per counter (and once per XCD / dimension for some): sum those
        for row in csv.DictReader(open(files[0])):
            k = (row["Kernel_Name"], str(row.get("Grid_Size", "")))
            per_dispatch.setdefault((k, row.get("Dispatch_Id", "")), {}).setdefault(row["Counter_Name"], 0.0)
            per_dispatch[(k, row.get("Dispatch_Id", ""))][row["Counter_Name"]] += float(row["Counter_Value"])
        for (k, _), cs in per_dispatch.items():
            for c, v in cs.items():
                agg.setdefault(k, {}).setdefault(c, []).append(v)
        traces = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
        if traces and si == 0:
            for row in csv.DictReader(open(traces[0])):
                if "Grid_Size" in row:
                    grid = str(row["Grid_Size"])
                else:
                    grid = str(int(row.get("Grid_Size_X", 1)) * int(row.get("Grid_Size_Y", 1)) * int(row.get("Grid_Size_Z", 1)))
                k = (row["Kernel_Name"], grid)
                agg.setdefault(k, {}).setdefault("duration_ns", []).append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    shutil.rmtree(tmp, ignore_errors=True)
    want = ("spmm_",) if what == "spmm" else ("filter_", "linear_", "gemm64", "slab_reduce", "select_topk", "score_gemm")
    out, lines = {}, []
    for (name, grid), cs in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("duration_ns", [0]))):
        if not any(w in name for w in want):
            continue
        clean = name.replace("(anonymous namespace)::", "").replace("void ", "")
        short = clean.split("(")[0].strip()
        rec = {c: sum(v) / len(v) for c, v in cs.items()}
        rec["calls"] = max(len(v) for v in cs.values())
        key = "%s grid=%s" % (short, grid)
        out[key] = rec
        lines.append(key + "   (calls %d)" % rec["calls"])
        for c in sorted(rec):
            if c != "calls":
                lines.append("   %-28s %16.0f" % (c, rec[c]))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in rec and rec.get("duration_ns"):
            # per-SIMD busy cycles / kernel cycles: the counter sums over the chip's 1024 SIMDs; 2.4 GHz nominal
            util = rec["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (rec["duration_ns"] * 2.4)
            lines.append("   %-28s %16.3f   (MFMA busy cycles per SIMD / (duration x 2.4 GHz))" % ("MfmaUtil_nominal_clock", util))
            rec["MfmaUtil_nominal_clock"] = util
        if "SQ_VALU_MFMA_BUSY_CYCLES" in rec and rec.get("SQ_BUSY_CU_CYCLES"):
            rec["MfmaUtil_busy_cu"] = rec["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * rec["SQ_BUSY_CU_CYCLES"])
            lines.append("   %-28s %16.3f   (MFMA busy cycles / (4 SIMDs x SQ_BUSY_CU_CYCLES))" % ("MfmaUtil_busy_cu", rec["MfmaUtil_busy_cu"]))
    os.makedirs(os.path.dirname(prefix), exist_ok=True)
    open(prefix + ".txt", "w").write("\n".join(lines) + "\n")
    json.dump(out, open(prefix + ".json", "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
