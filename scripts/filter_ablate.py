#!/usr/bin/env python3
"""What bounds a stage of the candidate filter (K4h)?  Runs the timing variant of flat_filter_kernel over a 10M x 768
index (bf16 rows / IP and f32 rows / COSINE, B = 256, k = 10) with pieces of the pipeline switched off
(VK_FILTER_ABLATE, see flat_filter_body): the answers are invalid, the times are the measurement.
    python scripts/filter_ablate.py [--rows N] [--dtypes bf16,f32] [--ablate 0,1,3,...]
Prints one JSON line per (dtype, ablation): kernel ms (HIP events of the library) and the phase counters' stderr line."""
import argparse, json, os, sys
# the experiment kernels exist only in the -DVK_EXPERIMENTS build of the library (csrc/Makefile `experiments`): build it and
# load it INSTEAD of libvkindex.so (the binding reads VKINDEX_LIB at import)
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if not os.environ.get("VKINDEX_LIB"):
    import subprocess
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(_ROOT, "valkey-search_amd", "csrc"), "experiments"])
    os.environ["VKINDEX_LIB"] = os.path.join(_ROOT, "valkey-search_amd", "libvkindex_exp.so")
os.environ.setdefault("VK_KERNEL_TIMING", "1")   # (HIP event pairs around the final pass: this script reads filter_kernel_ns)
if "--timing" in sys.argv:
    os.environ["VK_FILTER_TIMING"] = "1"     # (cycle counters per phase on stderr; the ticks themselves cost 15-25 %)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from _pkg import vsa
import bench as B


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--dtypes", default="bf16,f32")
    ap.add_argument("--ablate", default="-1,0,1,3,4,12,15,16,32,48,63")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--timing", action="store_true")
    ap.add_argument("--prio", default="0", help="VK_FILTER_PRIO values to run every ablation with")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    N, D, NB, K = args.rows, args.dim, 256, 10
    for dt in args.dtypes.split(","):
        bf = dt == "bf16"
        ix = vsa.Index("FLAT", D, "IP" if bf else "COSINE", initial_cap=N, device_id=0, dtype="bf16" if bf else "f32")
        base_ptr, stride = ix.device_rows(N)
        if bf:
            table = B.device_view_typed(base_ptr, (N, stride // 2), dev, "<i2").view(torch.bfloat16)
        else:
            table = B.device_view(base_ptr, (N, stride // 4), dev)
        A = None
        for lo, x in B.gen_rows(0, N, D, dev):
            table[lo: lo + x.shape[0], :D] = x
        torch.cuda.synchronize()
        ix.commit_device_rows(N, np.arange(N, dtype=np.uint64))
        g = torch.Generator(device=dev)
        g.manual_seed(1234)
        A = torch.randn(D, 32, generator=g, device=dev, dtype=torch.float32)
        Q = B.make_queries(A, NB, D, dev, 4242)
        od = torch.empty(NB, K, device=dev, dtype=torch.float32)
        ol = torch.empty(NB, K, device=dev, dtype=torch.int64)
        on = torch.empty(NB, device=dev, dtype=torch.int32)
        st = torch.cuda.Stream()
        for prio, abl in [(p, int(v)) for p in args.prio.split(",") for v in args.ablate.split(",")]:
            os.environ["VK_FILTER_PRIO"] = prio
            if abl < 0:
                os.environ.pop("VK_FILTER_ABLATE", None)      # the product kernel
            else:
                os.environ["VK_FILTER_ABLATE"] = str(abl)
            sys.stderr.write(f"[ablate] dtype {dt} ablate {abl}\n")
            sys.stderr.flush()
            with torch.cuda.stream(st):
                ix.search_batch_device(Q.data_ptr(), NB, K, od.data_ptr(), ol.data_ptr(), on.data_ptr(), stream=st.cuda_stream)
                torch.cuda.synchronize()
                s0 = ix.stats()
                for _ in range(args.steps):
                    ix.search_batch_device(Q.data_ptr(), NB, K, od.data_ptr(), ol.data_ptr(), on.data_ptr(), stream=st.cuda_stream)
                torch.cuda.synchronize()
                ix.search_batch_device(Q.data_ptr(), NB, K, od.data_ptr(), ol.data_ptr(), on.data_ptr(), stream=st.cuda_stream)
                torch.cuda.synchronize()
                s1 = ix.stats()
            nb = s1.filter_batches - s0.filter_batches
            ms = (s1.filter_kernel_ns - s0.filter_kernel_ns) / 1e6 / nb if nb else None
            print(json.dumps({"dtype": dt, "rows": N, "ablate": abl, "prio": int(prio), "filter_kernel_ms": ms, "launches": int(nb),
                              "survivors_per_query": round(s1.last_filter_candidates / NB, 1), "handed_over": int(s1.last_filter_fallback)}), flush=True)
        del ix, table
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
