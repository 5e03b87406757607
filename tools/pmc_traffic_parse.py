"""Turns the two `rocprofv3 --pmc` passes of tools/pmc_traffic.sh (FETCH_SIZE, WRITE_SIZE; counters only + kernel trace) into
profiles/pmc_traffic.json, the file bench.py reads `roofline.traffic` from.

  python tools/pmc_traffic_parse.py gpurun_out/pmc3 [--round r02]

Corrections (MI355X_MICROARCH.md "HBM"): both counters are in KiB (calibrated in round 1: torch's copy kernel writes 39.85 MB and
reports 38 912); on gfx950 FETCH_SIZE reports HALF of wide coalesced reads.  The factor is not assumed: it is calibrated in the same trace on torch's bf16 copy kernel (reads 4 B and writes 2 B per
element: FETCH/WRITE must come out at 2.0), and the calibration is written next to the result."""
import csv
import hashlib
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
csv.field_size_limit(1 << 30)


def per_kernel(path, counter):
    rows = {}
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                rows.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return rows


def main():
    d = Path(sys.argv[1])
    tag = sys.argv[sys.argv.index("--round") + 1] if "--round" in sys.argv else "r02"
    fetch = per_kernel(d / "FETCH_SIZE" / "pmc_counter_collection.csv", "FETCH_SIZE")
    write = per_kernel(d / "WRITE_SIZE" / "pmc_counter_collection.csv", "WRITE_SIZE")
    gemm = [k for k in fetch if "gemm_nt_kernel" in k or "gemm_nt_w4_kernel" in k]
    assert gemm, "no gemm_nt_kernel launch in the trace"
    dom = max(gemm, key=lambda k: sum(fetch[k]))
    red = [k for k in fetch if "gemm_hybrid_reduce" in k or "gemm_splitk_reduce" in k]
    copy = [k for k in fetch if "bfloat16_copy_kernel" in k and k in write]
    factor, calib = 2.0, None
    if copy:   # fp32 -> bf16 copy: reads 2x the bytes it writes
        k = max(copy, key=lambda k: max(fetch[k]))
        ratio = max(fetch[k]) / max(write[k])
        calib = {"kernel": "at::native::bfloat16_copy_kernel (fp32 -> bf16)", "FETCH_over_WRITE_raw": ratio, "expected": 2.0}
        factor = 2.0 / ratio
    n = len(fetch[dom])
    f_kb, w_kb = sum(fetch[dom]) / n, sum(write[dom]) / len(write[dom])
    m, nn, kk = (int(x) for x in (sys.argv[sys.argv.index("--shape") + 1: sys.argv.index("--shape") + 4] if "--shape" in sys.argv else (4864, 22016, 4096)))
    rec = {
        "bytes_per_launch": f_kb * 1024.0 * factor + w_kb * 1024.0,
        "fetch_bytes": f_kb * 1024.0 * factor, "write_bytes": w_kb * 1024.0, "fetch_correction": factor, "calibration": calib,
        "algorithmic_bytes": (m * kk + nn * kk + m * nn) * 2, "shape": [m, nn, kk], "launches": n, "kernel": dom[:120],
        "reduce_kernel": ({"fetch_bytes": sum(fetch[red[0]]) / len(fetch[red[0]]) * 1024.0 * factor, "write_bytes": sum(write[red[0]]) / len(write[red[0]]) * 1024.0}
                          if red and red[0] in write else None),
        "source": f"profiles/{tag}_pmc_traffic.json (tools/pmc_traffic.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, python tools/gemm_one.py {m} {nn} {kk})",
        "gemm_nt_sha16": hashlib.sha256((ROOT / "openvla-oft_amd" / "csrc" / "gemm_nt.hip").read_bytes()).hexdigest()[:16],
    }
    rec["ratio_to_algorithmic"] = rec["bytes_per_launch"] / rec["algorithmic_bytes"]
    for out in (ROOT / "profiles" / "pmc_traffic.json", ROOT / "profiles" / f"{tag}_pmc_traffic.json"):
        out.write_text(json.dumps(rec, indent=1) + "\n")
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
