"""profiles/<round>_ld_pmc.json from the per-kernel counter averages tools/pmc_ld.sh leaves in
gpurun_out/<tag>_pmc.json (the file bench.py reads for `roofline.valu`):

    python tools/pmc_ld_summary.py gpurun_out/r03pmc_pmc.json profiles/r03_ld_pmc.json <windows> <segments>

Derived quantities: instructions per (window, chunk of 64 individuals), the split of wave time, the VALU
issue cycles per SIMD against the kernel's own cycle count.  Cycles per VALU instruction: the (mask, count)
pairs of the segment loop -- 25 pairs per segment, v_and_b32 + s_nop 0 + v_bcnt_u32_b32 -- issue at the 3.2
cycles per instruction of tools/ubench/nop_mix.hip ("and NOP bcnt", 8 waves per SIMD), everything else at the
4.2 of the single-rate instructions (issue_rates.hip); the share of the pairs comes from the segment count
of the workload (PrepInfo n_segs / windows)."""
import json
import sys

src, dst = sys.argv[1], sys.argv[2]
# windows and segments of the workload: `config.windows_rank0` / `config.segments_rank0` of the bench line
n_win, n_chunks, n_segs = int(sys.argv[3]), 40, int(sys.argv[4])
raw = json.load(open(src))
# (the IBD1 form -- last template argument true -- where the timed steps reached it; else the form that counts everything)
names = [k for k in raw if "k_ld_popcount<" in k]
name = next((k for k in names if k.rstrip().endswith("true, true, true>")), names[0])
k = raw[name]
pairs = n_win * n_chunks
valu = k["SQ_INSTS_VALU"]
mfma = k.get("SQ_INSTS_MFMA", 0.0)
matrix_form = mfma > 0                         # k_ld_popcount<.., MX = true>: the counts of a haplotype word by one MFMA
if matrix_form:
    # SQ_INSTS_VALU counts the matrix instructions too.  The vector instructions of this form are a mix of the double-rate
    # class (v_and_b32, v_add_u32, v_bitop3_b32: 2.2-2.4 cycles alone) and the single-rate one (4.2-5: shifts, v_bcnt,
    # conversions, fp64): priced like the earlier rounds' mix, 3.5; a v_mfma_scale_f32_16x16x128_f8f6f4 (FP6 x FP4) holds the
    # SIMD's vector issue for its 16 cycles (tools/ubench/fp4_count.hip: times add)
    pair_instr = 0.0
    cyc = 3.5
    issue = ((valu - mfma) * cyc + mfma * 16.0) / 1024
else:
    pair_instr = 50.0 * n_segs * n_chunks          # 25 pairs x 2 instructions per (segment, chunk)
    cyc = (pair_instr * 3.2 + (valu - pair_instr) * 4.2) / valu
    issue = valu * cyc / 1024
kernel_cycles = k["GRBM_GUI_ACTIVE"] / 8
wc = k.get("SQ_WAVE_CYCLES")
out = {
    "source": "tools/pmc_ld.sh (rocprofv3 --kernel-trace --pmc passes over `bench.py --timed-only --steps 3 --warmup 1`; "
              "per-launch averages over the dispatches of each pass), summarised by tools/pmc_ld_summary.py",
    "kernel": name,
    "config": {"sites": 4000000, "n_ids": 2504, "window": 100, "targets": 1, "n_win": n_win, "n_chunks": n_chunks,
               "n_segs": n_segs},
    "per_launch": {c: v for c, v in k.items() if c != "dispatches_averaged"},
    "dispatches_averaged": k.get("dispatches_averaged"),
    "units": "SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md); "
             "GRBM_GUI_ACTIVE is summed over the 8 XCDs; VALUBusy / LdsUtil / SALUBusy are rocprofv3's derived percentages "
             "(VALUBusy prices every instruction at 4 cycles, so a stream with 2.2-cycle instructions reads above 100)",
    "derived": {
        "window_chunk_pairs": pairs,
        "valu_per_window_chunk": valu / pairs,
        "salu_per_window_chunk": k.get("SQ_INSTS_SALU", 0) / pairs,
        "lds_per_window_chunk": k.get("SQ_INSTS_LDS", 0) / pairs,
        "pair_instructions_per_window_chunk": pair_instr / pairs,
        "kernel_cycles": kernel_cycles,
        "mfma_per_window_chunk": mfma / pairs,
        "valu_issue_cycles_per_simd": issue,
        "valu_issue_share_of_kernel_cycles": issue / kernel_cycles,
    },
    "simds": 1024,
    "cycles_per_valu_instruction": {
        "value": cyc,
        "source": ("the matrix-core form's mix of double-rate and single-rate vector instructions priced at 3.5 like the earlier "
                   "rounds' mix; each of its matrix instructions at 16 cycles (tools/ubench/fp4_count.hip, profiles/r04_fp4_count.txt)")
                  if matrix_form else
                  "weighted: the segment loop's (mask, count) pairs at 3.2 cycles per instruction (profiles/r02_nop_mix.txt, "
                  "'and NOP bcnt'), all other vector instructions at 4.2 (profiles/r02_issue_rates.txt)"},
    "cycles_per_mfma": 16.0 if matrix_form else None,
    "nominal_clock_hz": 2.4e9,
}
if wc:
    out["derived"]["wave_time_split"] = {
        "parked_s_waitcnt": k["SQ_WAIT_ANY"] / wc,
        "issue_stall": k["SQ_WAIT_INST_ANY"] / wc,
        "issuing": 1.0 - (k["SQ_WAIT_ANY"] + k["SQ_WAIT_INST_ANY"]) / wc}
if "SQ_LDS_IDX_ACTIVE" in k and k["SQ_LDS_IDX_ACTIVE"]:
    out["derived"]["lds_bank_conflict_share_of_lds_cycles"] = k["SQ_LDS_BANK_CONFLICT"] / k["SQ_LDS_IDX_ACTIVE"]
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out["derived"], indent=1))
