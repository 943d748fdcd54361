#!/usr/bin/env python3
"""Where does the run-to-run spread of a scan launch come from?  One workload, many launches, each launch's kernel time (HIP
events on the launch stream) next to the GPU's clocks, power and temperature sampled in the same process (rocm-smi, every 20 ms).
usage: tools/launch_spread.py [--rows 5000 --sites 9000 --batch 1024 --coding] [--launches 200]  ->  one JSON object"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402


def sampler(stop, out):
    while not stop.is_set():
        t = time.perf_counter()
        try:
            p = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True, text=True, timeout=5)
            j = json.loads(p.stdout)
            card = next(iter(j.values()))
            num = lambda s: float("".join(c for c in str(s) if c.isdigit() or c == "."))     # noqa: E731
            rec = {"t": t}
            for k, v in card.items():
                kl = k.lower()
                if "sclk" in kl and "level" in kl:
                    rec["sclk_mhz"] = num(str(v).split("(")[-1])
                elif "mclk" in kl and "level" in kl:
                    rec["mclk_mhz"] = num(str(v).split("(")[-1])
                elif "fclk" in kl and "level" in kl:
                    rec["fclk_mhz"] = num(str(v).split("(")[-1])
                elif "power" in kl and "w" in kl and "package" in kl:
                    rec["power_w"] = num(v)
                elif "temperature" in kl and "hotspot" in kl or "temperature" in kl and "junction" in kl:
                    rec["temp_hot_c"] = num(v)
                elif "temperature" in kl and "mem" in kl:
                    rec["temp_mem_c"] = num(v)
            out.append(rec)
        except Exception as e:      # noqa: BLE001
            out.append({"t": t, "error": str(e)[:80]})
        time.sleep(0.02)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=5000)
    ap.add_argument("--sites", type=int, default=9000)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--coding", action="store_true")
    ap.add_argument("--launches", type=int, default=200)
    ap.add_argument("--pause-ms", type=float, default=0.0, help="idle time between launches")
    a = ap.parse_args()
    from polyfasta_amd import _lib
    _lib.init(0)
    w = bench.Workload(a.rows, a.sites, a.batch, coding=a.coding)
    samples, stop = [], threading.Event()
    th = threading.Thread(target=sampler, args=(stop, samples), daemon=True)
    th.start()
    for _ in range(3):
        w.plan.run(w.dev)
    _lib.check(_lib.load().pfa_sync())
    ms, at = [], []
    for i in range(a.launches):
        at.append(time.perf_counter())
        w.plan.run(w.dev)
        _lib.check(_lib.load().pfa_sync())
        ms.append(float(w.plan.kernel_ms(1)[0]))
        if a.pause_ms:
            time.sleep(a.pause_ms / 1e3)
    stop.set()
    th.join(timeout=6)
    ms = np.array(ms)
    good = [s for s in samples if "error" not in s]
    out = {"workload": w.label(), "kernel": w.plan.description, "launches": a.launches, "pause_ms": a.pause_ms,
           "kernel_ms": {"min": float(ms.min()), "p5": float(np.percentile(ms, 5)), "p50": float(np.percentile(ms, 50)),
                         "p95": float(np.percentile(ms, 95)), "max": float(ms.max()), "p95_over_p50": float(np.percentile(ms, 95) / np.percentile(ms, 50))},
           "frac_of_8TBs": {"best": w.alg_bytes / (ms.min() * 1e-3) / 8e12, "p50": w.alg_bytes / (np.percentile(ms, 50) * 1e-3) / 8e12,
                            "worst": w.alg_bytes / (ms.max() * 1e-3) / 8e12},
           "first_20_ms": [round(x, 3) for x in ms[:20].tolist()], "last_20_ms": [round(x, 3) for x in ms[-20:].tolist()],
           "drift_ms_first_to_last_quarter": float(ms[-len(ms) // 4:].mean() - ms[:len(ms) // 4].mean()),
           "smi_samples": len(good), "smi_errors": len(samples) - len(good)}
    if good:
        ts = np.array([s["t"] for s in good])
        for key in ("sclk_mhz", "mclk_mhz", "fclk_mhz", "power_w", "temp_hot_c", "temp_mem_c"):
            vals = np.array([s.get(key, np.nan) for s in good], dtype=float)
            if np.isfinite(vals).sum() < 3:
                continue
            out[key] = {"min": float(np.nanmin(vals)), "p50": float(np.nanmedian(vals)), "max": float(np.nanmax(vals))}
            # each launch against the nearest sample
            near = vals[np.clip(np.searchsorted(ts, np.array(at)), 0, len(vals) - 1)]
            ok = np.isfinite(near)
            if ok.sum() > 10 and np.nanstd(near[ok]) > 0:
                out[key]["corr_with_kernel_ms"] = float(np.corrcoef(near[ok], ms[ok])[0, 1])
    print(json.dumps(out))
    w.close()


if __name__ == "__main__":
    main()
