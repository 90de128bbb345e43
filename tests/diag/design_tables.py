"""Fills the @@...@@ placeholders of a DESIGN.md template from the measurement files of tests/diag/final_measure.sh.
    python tests/diag/design_tables.py <template> <dir with the json lines> > DESIGN.md
    python tests/diag/design_tables.py profiles/DESIGN_template.md profiles/round3_ > DESIGN.md     (the committed copies: a file-name prefix)"""
import json
import sys

tmpl, d = open(sys.argv[1]).read(), sys.argv[2]
prefix = not d.endswith("/") and d.endswith("_")  # profiles/roundN_<name> instead of <dir>/<name>


def line(name):
    return json.loads(open(f"{d}{name}.json" if prefix else f"{d}/{name}.json").read().strip().splitlines()[-1])


b50, b20, b1k, o1 = line("bench_line"), line("bench_line_steps20"), line("bench_line_batch1024"), line("bench_line_overlap1")
cpu = b50.get("cpu_baseline") or b20.get("cpu_baseline")
rows = [
    "| | round 1 | round 2 | **round 3** | CPU oracle (16-thread share) |",
    "|---|---|---|---|---|",
    f"| step (train + sample), default `--overlap 2`, 50 steps | 10.1 k jets/s (25.3 ms) | 12.1-12.3 k (20.9-21.1 ms) | **{b50['value'] / 1e3:.2f} k jets/s ({b50['ms_per_step']:.2f} ms)** | {cpu['value']:.1f} jets/s |",
    f"| the driver's `--steps 20 --warmup 5` | | 12.10 k (21.15 ms; `BENCH_r02.json`) | **{b20['value'] / 1e3:.2f} k ({b20['ms_per_step']:.2f} ms)** | |",
    f"| strictly sequential (`--overlap 1`) | 7.44 k (34.4 ms) | 8.33 k (30.7 ms) | {o1['value'] / 1e3:.2f} k ({o1['ms_per_step']:.1f} ms) | |",
    f"| yaml batch (`--batch 1024`) | 10.5 k (97.5 ms) | 12.5-12.6 k (81.3-81.8 ms) | {b1k['value'] / 1e3:.2f} k ({b1k['ms_per_step']:.1f} ms) | |",
    f"| 100-step sample alone, 256 jets | 33.4-33.6 ms | 29.2-30.3 ms | {b50['roofline']['kernel_alone_ms']:.1f} ms (HIP events, 3 launches alone); {b50['sample_ms']:.1f} ms inside the timed region | {cpu['sample_s_scaled']:.1f} s |",
    f"| train step alone, 256 jets | 1.11 ms | 0.80-0.83 ms | **{b50['train_ms_alone']:.2f} ms** | {cpu['train_ms_median']:.0f} ms |",
    f"| train step alone, 1024 jets | 3.2 ms | 2.37-2.39 ms | **{b1k['train_ms_alone']:.2f} ms** | |",
    f"| `roofline.frac` (executed FLOP / timed wall time / 157.3 TFLOP/s) | 0.54-0.55 | 0.65-0.66 | **{b20['roofline']['frac']:.3f}** (steps 20) / {b50['roofline']['frac']:.3f} (50) / {b1k['roofline']['frac']:.3f} (batch 1024) | |",
    f"| `roofline.mfma_busy` (SQ counter, one launch alone) | | (missing) | **{b50['roofline']['mfma_busy']:.3f}** | |",
    f"| same sample, bf16 operands (informational, not `value`) | 13.9 ms | 11.1-12.7 ms | {b50['bf16_mfma_sampler']['sample_ms']:.2f} ms (max dev. from fp32 {b50['bf16_mfma_sampler']['max_abs_dev_from_f32']:.1e}) | |",
    f"| same sample, split-fp16 operands (informational) | 18.5 ms | 18.3 ms | {b50['f16x3_mfma_sampler']['sample_ms']:.2f} ms (max dev. {b50['f16x3_mfma_sampler']['max_abs_dev_from_f32']:.1e}) | |",
]
tmpl = tmpl.replace("@@HEADLINE_TABLE@@", "\n".join(rows))
tmpl = tmpl.replace("@@MS20@@", f"{b20['ms_per_step']:.2f}")
tmpl = tmpl.replace("@@BUSY_WALL@@", f"{33.70e9 / (1024 * b20['ms_per_step'] * 1e-3 * 2.29e9):.2f}")
sec = [json.loads(l) for l in open(f"{d}secondary_bench_lines.jsonl" if prefix else f"{d}/secondary_lines.jsonl") if l.strip().startswith("{")]
prev = {("cfg 2", "f32"): ("21.6 k", "34.5 k"), ("cfg 2", "bf16"): ("27.6 k", "51.9 k"), ("cfg 4", "f32"): ("633", "720"),
        ("cfg 5", "f32"): ("463", "560"), ("SURVEY", "f32"): ("706", "901")}
rows = ["| BASELINE config | round 1 | round 2 | **round 3** | CPU oracle | `roofline.frac` (executed) |", "|---|---|---|---|---|---|"]
for e in sec:
    w = e["config"]["workload"]
    key = (w.split(":")[0].split()[0] + (" " + w.split(":")[0].split()[1] if w.startswith("cfg") else ""), e["dtype"])
    p = prev.get((key[0] if key[0] != "SURVEY 8(f)-4" else "SURVEY", key[1]), prev.get(("SURVEY", "f32")) if w.startswith("SURVEY") else ("", ""))
    peak = "bf16 peak 2.5 PF" if e["dtype"] == "bf16" else "fp32 peak"
    rows.append(f"| {w.split(',')[0][:70]} ({e['dtype']}) | {p[0]} | {p[1]} | **{e['value']:,.0f} jets/s** ({e['ms_per_step']:.1f} ms / step) | "
                f"{e['cpu_baseline']['value']:.2f} jets/s | {e['roofline']['frac']:.3f} of the {peak} |")
tmpl = tmpl.replace("@@SECONDARY_TABLE@@", "\n".join(rows))
sys.stdout.write(tmpl)
