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


class ClockSampler:
    """engine clock (MHz) and socket power (W) of GPU 0 while a loop of launches runs: sysfs hwmon where it is there (a read per
    millisecond), else `rocm-smi --showclocks --showpower` as fast as it answers"""

    def __init__(self):
        import ctypes, glob
        # the hwmon node of THE device the kernels run on (a box holds eight cards): by PCI bus id of HIP device 0
        bus = None
        try:
            buf = ctypes.create_string_buffer(64)
            hip = ctypes.CDLL("libamdhip64.so")
            if hip.hipDeviceGetPCIBusId(buf, 64, 0) == 0:
                bus = buf.value.decode().lower()
        except OSError:
            pass
        base = f"/sys/bus/pci/devices/{bus}/hwmon/hwmon*/" if bus else "/sys/class/drm/card*/device/hwmon/hwmon*/"
        self.freq = (glob.glob(base + "freq1_input") or [None])[0]
        pw = glob.glob(base + "power1_average") + glob.glob(base + "power1_input")
        self.power = pw[0] if pw else None
        self.source = f"sysfs hwmon of {bus}" if self.freq else "rocm-smi"
        self.mhz, self.watts, self._stop = [], [], False

    def _once(self):
        if self.freq:
            try:
                self.mhz.append(int(open(self.freq).read()) / 1e6)
                if self.power:
                    self.watts.append(int(open(self.power).read()) / 1e6)
            except (OSError, ValueError):
                pass
            return
        import re, subprocess
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        m = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
        if m:
            self.mhz.append(float(m.group(1)))
        m = re.search(r"Socket (?:Graphics Package )?Power \(W\): ([0-9.]+)", out)
        if m:
            self.watts.append(float(m.group(1)))

    def run(self, fn):
        import threading, time

        def loop():
            while not self._stop:
                self._once()
                if self.freq:
                    time.sleep(0.001)
        self.mhz, self.watts, self._stop = [], [], False
        t = threading.Thread(target=loop)
        t.start()
        try:
            fn()
        finally:
            self._stop = True
            t.join()
        q = lambda v, f: round(sorted(v)[min(len(v) - 1, int(f * len(v)))], 1) if v else None
        return {"source": self.source, "samples": len(self.mhz), "sclk_mhz_p10": q(self.mhz, 0.1), "sclk_mhz_p50": q(self.mhz, 0.5),
                "sclk_mhz_p90": q(self.mhz, 0.9), "power_w_p50": q(self.watts, 0.5), "power_w_max": max(self.watts) if self.watts else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sample-clock", action="store_true", help="engine clock and socket power sampled while each variant's loop runs")
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--dtypes", default="bf16,f32")
    ap.add_argument("--ablate", default="-1,0,1,3,4,12,15,16,32,48,63")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--timing", action="store_true")
    ap.add_argument("--prio", default="0", help="VK_FILTER_PRIO values to run every ablation with")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    sampler = ClockSampler() if args.sample_clock else None
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

                def loop():
                    for i in range(args.steps):
                        ix.search_batch_device(Q.data_ptr(), NB, K, od.data_ptr(), ol.data_ptr(), on.data_ptr(), stream=st.cuda_stream)
                        if i % 16 == 15:   # (the library keeps 32 event pairs per context: none may be reused before it completed)
                            torch.cuda.synchronize()
                    torch.cuda.synchronize()
                clock = sampler.run(loop) if sampler else (loop(), None)[1]
                ix.search_batch_device(Q.data_ptr(), NB, K, od.data_ptr(), ol.data_ptr(), on.data_ptr(), stream=st.cuda_stream)
                torch.cuda.synchronize()
                s1 = ix.stats()
            nb = s1.filter_batches - s0.filter_batches
            ms = (s1.filter_kernel_ns - s0.filter_kernel_ns) / 1e6 / nb if nb else None
            print(json.dumps({"dtype": dt, "rows": N, "ablate": abl, "prio": int(prio), "filter_kernel_ms": ms, "launches": int(nb),
                              "survivors_per_query": round(s1.last_filter_candidates / NB, 1), "handed_over": int(s1.last_filter_fallback),
                              "clock": clock}), flush=True)
        del ix, table
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
