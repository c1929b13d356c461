"""Turns the counter summaries of the decode / SampleRNN-generation PMC sessions (tools/gpu_session.sh pmcdecode / pmcsr:
FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes over tools/pm_timing.py MODE=decode / tools/sr_timing.py) into
profiles/rNN_pmc_secondary.json: HBM-side bytes per decode step (pm_kernel: one launch = the whole decode loop) and per
SampleRNN sample step (srp_kernel: one launch = the FRAME_SIZE sample steps between two frame-tier steps).

    python tools/pmc_secondary.py <decode summary.txt | -> <sr summary.txt | -> <out.json> [session label]

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half of the bytes of wide coalesced reads, so it
is doubled; both counters are in KiB.  bench.py reports the figures as counter_bytes_per_step / frac_counter and refuses
them once the sources of the kernels have changed (secondary_source_digest)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import secondary_source_digest  # noqa: E402


def kernel_counters(path, needle):
    """{counter: (average per launch, launches)} of the kernel whose name contains `needle` (pmc_summary.py format)."""
    out, cur = {}, None
    for line in open(path):
        if not line.startswith(" "):
            cur = line.strip()
            continue
        m = re.match(r"\s+(\w+)\s+([0-9.eE+-]+)\s+\(x(\d+)\)", line)
        if m and cur and needle in cur:
            out[m.group(1)] = (float(m.group(2)), int(m.group(3)))
    return out


dec, sr, outp = sys.argv[1], sys.argv[2], sys.argv[3]
blob = {"recipe": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024"}
if dec != "-":
    c = kernel_counters(dec, "pm_kernel")
    steps = int(os.environ.get("DECODE_STEPS", "1000"))  # tools/pm_timing.py MODE=decode: S = 1000 frames per launch
    fetch, write = 2.0 * c["FETCH_SIZE"][0] * 1024, c["WRITE_SIZE"][0] * 1024
    blob["decode_cfg3"] = {"kernel": "pm_kernel", "steps_per_launch": steps, "launches": c["FETCH_SIZE"][1],
                           "fetch_bytes_per_step": round(fetch / steps), "write_bytes_per_step": round(write / steps),
                           "bytes_per_step": round((fetch + write) / steps)}
if sr != "-":
    c = kernel_counters(sr, "srp_kernel")
    steps = 10  # FRAME_SIZE sample steps per launch
    fetch, write = 2.0 * c["FETCH_SIZE"][0] * 1024, c["WRITE_SIZE"][0] * 1024
    blob["samplernn_cfg5"] = {"kernel": "srp_kernel", "steps_per_launch": steps, "launches": c["FETCH_SIZE"][1],
                              "fetch_bytes_per_step": round(fetch / steps), "write_bytes_per_step": round(write / steps),
                              "bytes_per_step": round((fetch + write) / steps),
                              "note": "the sample kernel only: the frame / big-frame tier launches between two of its launches are not in this figure"}
blob["source_digest"] = secondary_source_digest(ROOT)
if len(sys.argv) > 4:
    blob["session"] = sys.argv[4]
json.dump(blob, open(outp, "w"), indent=1)
print(json.dumps(blob))
