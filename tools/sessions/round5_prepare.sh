#!/bin/bash
# Round 5, BEFORE the first GPU call (CPU side, ~2 min per build): the A/B builds of k_resolve that round 4 left compiled and ISA-checked
# but unmeasured (its GPU minutes were spent).  Every option is compile-time and OFF in the product build.
#   SW_RESOLVE_SPEC    the receiver's view of the replica's hot subject (watch slot 0's), and the first two entries of its queue, are fetched in
#                      the SAME round trip as its inbox line instead of in a second one (ISA: four loads issued ~145 instructions ahead of the
#                      line's, no vmcnt wait between; 128 VGPR / 32 B scratch).  A receiver's chain: line -> {queue, view} -> apply becomes
#                      {line, queue, view} -> apply.  Results cannot depend on the guess (a wrong one is looked up as before).
#   SW_RESOLVE_LINE1 + SW_RESOLVE_WAVES=5
#                      16 instead of 64 bytes of the inbox line parked in LDS (37 -> 25 KB per workgroup) and five waves per SIMD
#                      (96 VGPR / 128 B scratch): five workgroups per CU where LDS and registers both allowed four.
#   SW_MASS_HBMQ       handles with the dense pair store (config #4) edit the memberlist queue in HBM (slot-major: a wave touches 1 KB per slot)
#                      instead of staging queue_cap x 256 x 16 B in LDS: 21 KB of LDS per workgroup instead of 149 KB at queue_cap 32, FOUR
#                      workgroups of k_resolve<MASS> per CU instead of ONE (128 VGPR, no scratch: the compiler's report).  Bit-identical to the
#                      checker on the emulator (tools/emu: the dense-store tests, serf intents in rows).  The config-4 leg is the measure.
# All four options have run the parity tests on the emulated kernels (tools/emu/README.md): the first GPU call measures, it does not debug.
# After ANY kernel change of round 5, before it costs a GPU minute:  tools/emu/build.sh && python -m pytest tests/test_emulated_kernels.py
# (1.5 min; the change against the checker on the host), then  SWIMSIM_EMU_SO=tools/emu/_build/libswimsim_emu.so python -m pytest tests -m gpu -k <what it touches>
# and, for anything that adds a barrier-free hand-over between waves or workgroups,  tools/emu/sweep.sh race.
cd "$(dirname "$0")/../.."
mkdir -p _ab
bash tools/build_variant.sh _ab/lib_0ref.so &
bash tools/build_variant.sh _ab/lib_spec.so -DSW_RESOLVE_SPEC &
bash tools/build_variant.sh _ab/lib_line1_w5.so -DSW_RESOLVE_LINE1 -DSW_RESOLVE_WAVES=5 &
bash tools/build_variant.sh _ab/lib_spec_line1_w5.so -DSW_RESOLVE_SPEC -DSW_RESOLVE_LINE1 -DSW_RESOLVE_WAVES=5 &
bash tools/build_variant.sh _ab/lib_hbmq.so -DSW_MASS_HBMQ &
wait
ls -la _ab/
