"""Static VALU instruction-class mix of the VALU-bound kernels (from the gfx950 assembly hipcc emits for their translation units),
for the pipe-level roofline statement VERDICT r3 asked for: issue_frac = sum_class(wave-instructions x cycles) / (SIMDs x clock x time).
Run in the build container:  python scripts/isa_class_mix.py  ->  profiles/r04_isa_class_mix.json

Issue cost per wave-instruction (wave64 on CDNA4's SIMD-32; MI355X_MICROARCH.md "Per-instruction cycle constants" where measured,
otherwise the datasheet rate): fp32 / integer / moves 2 cycles (v_fma_f32: 2, measured); packed fp32 4 (the fp32 peak of 64
flop/clk/SIMD is already reached by plain v_fma_f32, so a packed instruction is two of them); fp64 arithmetic, compares and
conversions 4 (78.6 TFLOP/s = 32 flop/clk/SIMD); fp32 transcendentals 8 (quarter rate); fp64 transcendentals 16 (quarter rate);
v_mfma_f64_16x16x4_f64 64 (2048 flop at 32 flop/clk).  The mix is STATIC over the kernel's text: exact for straight-line bodies,
an approximation where prologue / epilogue code sits next to the hot loop - used only as weights for the dynamic SQ_INSTS_VALU count."""
import json
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "baybe_amd" / "csrc"
TARGETS = {  # translation unit -> (substring of the mangled name, the name rocprofv3 prints)
    "bbh_acq.hip": [("bbh_qlogei_pending_q_kernelILi2", "bbh_qlogei_pending_q_kernel<2>"), ("bbh_qlogei_pending_q_kernelILi3", "bbh_qlogei_pending_q_kernel<3>"),
                    ("bbh_qlogei_pending_q_kernelILi4", "bbh_qlogei_pending_q_kernel<4>"), ("bbh_qlogei_pending_q_kernelILi5", "bbh_qlogei_pending_q_kernel<5>"),
                    ("bbh_qlognehvi_lin_kernelILi3ELb1", "bbh_qlognehvi_lin_kernel<3, true>")],
    "bbh_select.hip": [("bbh_qlogei_q1s_kernel", "bbh_qlogei_q1s_kernel"), ("bbh_select_kernel", "bbh_select_kernel")],
    "bbh_fused_small_b.hip": [("bbh_small_posterior_kernelILi4ELi0ELi4", "bbh_small_posterior_kernel<4, 0, 4>")],
    "bbh_fused_small_a.hip": [("bbh_small_posterior_kernelILi2ELi0ELi2", "bbh_small_posterior_kernel<2, 0, 2>")],
}
CYCLES = {"fp32_int": 2, "pk_f32": 4, "f64": 4, "trans_f32": 8, "trans_f64": 16, "mfma_f64": 64}
TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")


def classify(op: str):
    if not op.startswith("v_"):
        return None
    if op.startswith("v_mfma"):
        return "mfma_f64"
    if op.startswith(TRANS):
        return "trans_f64" if "f64" in op else "trans_f32"
    if op.startswith("v_pk_"):
        return "pk_f32"
    if "f64" in op or op.endswith("_b64") and op.startswith(("v_lshl", "v_lshr", "v_ashr")):
        return "f64"
    return "fp32_int"


def main():
    out = {"cycles_per_wave_instruction": CYCLES, "kernels": {}}
    for tu, names in TARGETS.items():
        asm = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", "-",
                              str(CSRC / tu)], capture_output=True, text=True, cwd=CSRC).stdout
        blocks = re.split(r"\n(?=_Z[\w]+:)", asm)
        for name, shown in names:
            body = next((b for b in blocks if b.startswith("_Z") and name in b.split(":", 1)[0]), None)
            if body is None:
                print("not found:", name, file=sys.stderr)
                continue
            body = body.split("s_endpgm")[0]
            mix = {k: 0 for k in CYCLES}
            for line in body.splitlines():
                m = re.match(r"\s+(v_\w+)", line)
                if m:
                    c = classify(m.group(1))
                    if c:
                        mix[c] += 1
            valu = sum(v for k, v in mix.items() if k != "mfma_f64")
            out["kernels"][shown] = {
                "mangled": body.split(":", 1)[0],
                "static_counts": mix,
                "valu_instructions": valu,
                "mean_cycles_per_valu_instruction": sum(mix[k] * CYCLES[k] for k in mix if k != "mfma_f64") / max(valu, 1),
            }
    dst = ROOT / "profiles" / "r04_isa_class_mix.json"
    dst.write_text(json.dumps(out, indent=1))
    for k, v in out["kernels"].items():
        print(k[:70], v["static_counts"], round(v["mean_cycles_per_valu_instruction"], 2))


if __name__ == "__main__":
    main()
