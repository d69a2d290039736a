#!/usr/bin/env python3
"""Config 5 through the plugin API on one MI355X: ms per training step of the plain FREEDOM plugin (hipGraph replay), of the
feature-sliced plugin on a one-rank RCCL group with its collectives forced (eager, and replayed as a hipGraph: config
`dist_graph_step`) and of the row-sharded plugin (eager) -- what the multi-GPU code paths cost before any GPU is added
(round-3 review: the forced-collective single-rank step was 5.10 ms against 3.66 ms plain).  Dataset built in memory
(tests/test_c5_e2e_gpu.py).      python tools/c5_sliced_step.py [steps]"""
import os
import socket
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    import torch.distributed as dist
    from mmrec_amd.common.trainer import Trainer
    from tests.test_c5_e2e_gpu import build_c5
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(s.getsockname()[1]))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    s.close()
    root = tempfile.mkdtemp(prefix="mmrec_c5s_", dir="/tmp")
    runs = [("plain FREEDOM (hipGraph replay)", False, {}),
            ("SlicedFREEDOM, 1 rank, collectives forced, eager", True, {"dist_layout": "dslice"}),
            ("SlicedFREEDOM, 1 rank, collectives forced, hipGraph replay", True, {"dist_layout": "dslice", "dist_graph_step": True}),
            ("RowShardedFREEDOM, 1 rank, collectives forced, eager", True, {"dist_layout": "rows"})]
    only = os.environ.get("MMREC_C5S_ONLY")
    for name, sharded, hyper in runs:
        if only and only not in name:
            continue
        if sharded and not dist.is_initialized():
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", torch.cuda.current_device()))
        config, train_data, _, model = build_c5(root, sharded=sharded, hyper=hyper)
        trainer = Trainer(config, model)
        keep = torch.multinomial(model.edge_values, int(model.edge_values.numel() * 0.2),
                                 generator=torch.Generator(device=model.edge_values.device).manual_seed(5))
        model.set_kept_edges(keep)
        batches = []
        for b in train_data:
            batches.append(b)
            if len(batches) == steps + 5:
                break
        trainer._train_epoch(batches[:5], 0)
        torch.cuda.synchronize()
        t0 = time.time()
        _, per = trainer._train_epoch(batches[5:], 0)
        torch.cuda.synchronize()
        ms = (time.time() - t0) / steps * 1e3
        graphed = getattr(trainer, "_graphed", None)
        state = "no graph" if graphed is None else ("capture FAILED -> eager" if graphed.failed else
                                                    ("replayed" if graphed.graph is not None else "not captured"))
        print("[c5-sliced] %-62s %.2f ms/step (%s; %s; loss %.6f -> %.6f)" %
              (name, ms, type(model).__name__, state, float(per[0]), float(per[-1])), flush=True)
        del trainer, model, train_data, batches
        torch.cuda.empty_cache()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
