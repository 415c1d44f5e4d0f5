"""Register, scratch and occupancy budget of the tick kernels, from the compiler's own report (hipcc cross-compiles gfx950
without a GPU).  What the design rests on — k_begin at five waves per SIMD, k_resolve (one wave per node block since round 5) at four without a spilled context, the
others light — is easy to lose with one more statement in a handler (a context that fell out of registers made
k_resolve twice as slow); this test says so on the CPU box, before anything is measured."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# kernel (mangled-name fragment) -> (max VGPRs, max scratch bytes per lane, min waves per SIMD)
BUDGET = {
    "k_beginILi4ELb0ELb0ELb0ELb0E": (96, 16, 5),   # the bench's instantiation: fan-out <= 4, no serf events, one shard, no dense pair store, no tile buckets
    "k_beginILi4ELb0ELb0ELb0ELb1E": (96, 48, 5),   # ... with tile buckets (SWIMSIM_TILEBUCKETS=1; its carry role walks the areas twice)
    "k_beginILi4ELb0ELb0ELb1ELb0E": (96, 32, 5),   # ... with the dense pair store (config #4 / #5 legs)
    "k_beginILi8ELb1ELb1ELb1ELb0E": (96, 128, 5),  # the heaviest one (fan-out 8, serf events, sharded, dense store)
    "9k_deliverILb0ELb0EE": (64, 0, 7),
    "9k_deliverILb0ELb1EE": (96, 0, 5),        # tile buckets: the per-tile drain, four records per lane in flight
    "9k_resolveILb0ELb0ELb0EE": (128, 16, 4),  # the bench's instantiation: no dense store, no serf event layer, fixed membership — nothing of those three compiled in
    "9k_resolveILb1ELb0ELb0EE": (128, 32, 4),  # ... the dense pair store (config #4's leg)
    "9k_resolveILb0ELb0ELb1EE": (128, 48, 4),  # ... membership that changes (per-observer estNumNodes, the f64 suspicion formula)
    "9k_resolveILb0ELb1ELb0EE": (128, 128, 4), # ... serf's event layer: intents with their statusLTime ordering, the event buffer (cold paths that spill)
    "9k_resolveILb1ELb1ELb1EE": (128, 144, 4), # ... all three (config #5's leg with joins)
    "8k_censusPK": (32, 0, 8),
    "8k_finishPK": (64, 0, 8),
    "15k_census_finishPK": (48, 0, 8),       # round 5: the recount and the tick's epilogue in one launch
    "17k_inbox_sort_hugePK": (48, 0, 8),     # inboxes beyond what LDS sorts (config #4's recovery)
    "16k_reconnect_scanPK": (32, 0, 8),      # serf reconnect over the dense store: a wave per (due node, chunk of rows)
    "7k_quietPK": (40, 0, 8),
    "14k_coord_updatePK": (128, 0, 4),
    "13k_expire_massPK": (64, 0, 8),
    "12k_inbox_sortPK": (48, 0, 8),           # big inboxes, a workgroup each, in LDS
    "16k_inbox_sort_medPK": (48, 0, 8),       # inboxes of 12-127 messages, a wave each, in LDS
}


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="no hipcc here")
def test_tick_kernels_stay_within_their_register_budget(tmp_path):
    out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-Wno-unused-value", "--cuda-device-only", "-c",
                          "-Rpass-analysis=kernel-resource-usage", "-o", str(tmp_path / "dev.o"), os.path.join(ROOT, "consul_amd", "csrc", "swim_host.hip")],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    rep = {}
    cur = None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = rep.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+(VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).split()[0]] = int(m.group(2))
    for frag, (vgpr, scratch, waves) in BUDGET.items():
        hits = {k: v for k, v in rep.items() if frag in k}
        assert len(hits) == 1, (frag, list(hits))
        name, r = next(iter(hits.items()))
        assert r["VGPRs"] <= vgpr and r["ScratchSize"] <= scratch and r["Occupancy"] >= waves, (name, r, (vgpr, scratch, waves))
