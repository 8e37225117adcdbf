"""The register-staged pipelined kernels issue their global loads from inline asm; hipcc does not track those, so it
could copy, spill or read a destination register before the data has landed (tools/audit_asm_loads.py).  This test
cross-compiles the two sources to gfx950 assembly (no GPU needed) and requires a CLEAN audit -- zero touches of an
in-flight destination on any control-flow path -- for EVERY built instantiation that uses the protocol: a guard
against a compiler or code change breaking it silently.  (Round 3: the weight-gradient pipeline retires both register
sets before its band ends, the stem forward walks its rows in pairs without a mid-loop exit; both were findings before.)"""
import importlib.util
import re
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pytorch-face-detection-from-scratch_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

KERNELS = {
    # every instantiation of the register-staged pipelined weight gradient that is still built: <VW, DBG, LPR, PK4> =
    # 60x60 (float4 lanes), 15x15 and 30x30 (float4 quads of narrow rows), the 32-lane one-float form; the 16-lane
    # one-float form is no longer a pipeline (plan_x3 routes those tiny maps to the staged kernel)
    "fdet_wgrad3x3_x3.hip": ["_ZN12_GLOBAL__N_118k_wgrad3x3_x3_pipeILi4ELi0ELi16ELi0E", "_ZN12_GLOBAL__N_118k_wgrad3x3_x3_pipeILi4ELi0ELi16ELi1E",
                             "_ZN12_GLOBAL__N_118k_wgrad3x3_x3_pipeILi4ELi0ELi32ELi1E", "_ZN12_GLOBAL__N_118k_wgrad3x3_x3_pipeILi1ELi0ELi32ELi0E"],
    # k_stem_fwd_x3_pipe<PSO, P16, U8>: fp32 output, PS output, its precision16 form, and the two uint8-frame forms
    "fdet_stem_x3.hip": ["_ZN12_GLOBAL__N_118k_stem_fwd_x3_pipeILb0ELb0ELb0E", "_ZN12_GLOBAL__N_118k_stem_fwd_x3_pipeILb1ELb0ELb0E",
                         "_ZN12_GLOBAL__N_118k_stem_fwd_x3_pipeILb1ELb1ELb0E", "_ZN12_GLOBAL__N_118k_stem_fwd_x3_pipeILb1ELb0ELb1E",
                         "_ZN12_GLOBAL__N_118k_stem_fwd_x3_pipeILb1ELb1ELb1E", "_ZN12_GLOBAL__N_120k_stem_wgrad_x3_pipeILb0E",
                         "_ZN12_GLOBAL__N_120k_stem_wgrad_x3_pipeILb1E"],
}
# The PS kernels issue LDS-DMA from asm (no register destination); their hand-counted waits are audited further down.


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("src", sorted(KERNELS))
def test_no_instruction_touches_in_flight_asm_loads(src, tmp_path):
    spec = importlib.util.spec_from_file_location("audit_asm_loads", os.path.join(ROOT, "tools", "audit_asm_loads.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    out = tmp_path / (src + ".s")
    subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                    "-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", str(out)], check=True, capture_output=True, timeout=600)
    for k in KERNELS[src]:
        nload, bad, foreign = mod.audit(k, str(out), verbose=False)
        assert nload > 0, f"{k}: no asm buffer loads found (kernel renamed?)"
        print(f"{k}: {nload} asm loads, {bad} touches of in-flight destinations, {foreign} compiler-issued vector-memory ops beside them")
        assert bad == 0, f"{k}: {bad} instructions touch an in-flight asm load destination: {mod.audit.last_touches[:4]}"
    # no pipelined instantiation escapes the list above
    import re
    built = set(re.findall(r"^(_ZN12_GLOBAL__N_1\d+k_(?:wgrad3x3_x3_pipe|stem_fwd_x3_pipe|stem_wgrad_x3_pipe)\w*):", open(out).read(), re.M))
    assert built and all(any(b.startswith(k) for k in KERNELS[src]) for b in built), built


# ---------------------------------------------------------------------------------------------------------------------
# The LDS-DMA kernels (fdet_conv3x3_ps.hip: 8 instantiations, fdet_wgrad3x3_ps.hip, fdet_chain_x3.hip): asm pieces have no
# register destination, but the waits that open an LDS buffer are `s_waitcnt vmcnt(K)` with K a HAND COUNT of
# compiler-issued stores / prefetch loads.  tools/audit_vmcnt.py proves from the cross-compiled assembly that every K is
# backed by a run of at least K vector-memory operations behind the nearest DMA piece and that no other run length can
# reach the wait (a merged, dropped or duplicated store changes a run length).
PS_CONV_TUS = list(range(16))          # PS_TU = 8 * P16 + 2 * mode + (WP == 64): FWD_FULL, DGRAD_ACT, FWD_POOL, DGRAD_ADDPOOL x {32, 64} x precision
VM_KERNELS = {
    # k_wgrad3x3_ps<WP, FL1 = true> (the default one-band flight) waits with vmcnt(0) only: listed so that a future counted
    # wait is audited.  FL1 = false counts asm DMA pieces themselves (volatile asm: the compiler cannot merge or drop them).
    "fdet_wgrad3x3_ps.hip": ["_ZN12_GLOBAL__N_113k_wgrad3x3_psILi64ELb1ELb0E", "_ZN12_GLOBAL__N_113k_wgrad3x3_psILi32ELb1ELb0E",
                             "_ZN12_GLOBAL__N_113k_wgrad3x3_psILi16ELb1ELb0E", "_ZN12_GLOBAL__N_113k_wgrad3x3_psILi64ELb1ELb1E",
                             "_ZN12_GLOBAL__N_113k_wgrad3x3_psILi32ELb1ELb1E", "_ZN12_GLOBAL__N_113k_wgrad3x3_psILi16ELb1ELb1E"],
    "fdet_chain_x3.hip": ["_ZN12_GLOBAL__N_116k_block_chain_psILb0ELb0E", "_ZN12_GLOBAL__N_116k_block_chain_psILb1ELb0E",
                          "_ZN12_GLOBAL__N_116k_block_chain_psILb0ELb1E", "_ZN12_GLOBAL__N_116k_block_chain_psILb1ELb1E",
                          "_ZN12_GLOBAL__N_116k_block_chain_x3ILb0E"],
}


def _compile_s(args):
    src, out, defs = args
    subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                    "-S", "--cuda-device-only"] + defs + [os.path.join(CSRC, src), "-o", out], check=True, capture_output=True, timeout=1200)
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_hand_counted_vmcnt_waits_of_the_lds_dma_kernels(tmp_path):
    from concurrent.futures import ThreadPoolExecutor
    spec = importlib.util.spec_from_file_location("audit_vmcnt", os.path.join(ROOT, "tools", "audit_vmcnt.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    jobs = [("fdet_conv3x3_ps.hip", str(tmp_path / f"ps{tu}.s"), [f"-DPS_TU={tu}"]) for tu in PS_CONV_TUS]
    jobs += [(src, str(tmp_path / (src + ".s")), []) for src in VM_KERNELS]
    with ThreadPoolExecutor(max_workers=8) as ex:
        outs = list(ex.map(_compile_s, jobs))
    # The fused pooled forward (mode 2: TUs 4, 5, 12, 13) is checked with rule B only.  Its K counts the epilogue's STORES;
    # between the DMA and the wait every path also carries the epilogue's >= 32 skip-tensor loads (uncounted slack: K <= 44
    # <= 32 + stores), and hipcc compiles each `offset = valid ? address : out-of-range` store as TWO stores under
    # complementary EXEC masks, each behind its own s_cbranch_execz -- at least one of the pair always issues, which can
    # only add operations, but a walk that cannot pair the masks sees every subset as a path (rule A would flag those).
    todo = [("_ZN12_GLOBAL__N_112k_conv3x3_ps", o, (tu & 7) // 2 != 2) for tu, o in zip(PS_CONV_TUS, outs[:len(PS_CONV_TUS)])]
    for src, o in zip(VM_KERNELS, outs[len(PS_CONV_TUS):]):
        todo += [(k, o, True) for k in VM_KERNELS[src]]
    n_waits = 0
    for k, o, rule_a in todo:
        res, ndma = mod.audit(k, o)
        assert ndma > 0, f"{k}: no LDS-DMA pieces found (kernel renamed?)"
        bad = mod.check(res, rule_a=rule_a)
        print(f"{k} ({os.path.basename(o)}): {ndma} DMA pieces, {len(res)} hand-counted waits: {sorted({K for _, K, _ in res})}")
        assert not bad, f"{k} ({os.path.basename(o)}): {bad[:3]}"
        n_waits += len(res)
    assert n_waits >= 40                                   # the conv instantiations alone hold ~60 counted waits
    # every DMA piece overwrites M0 without declaring it (fdet_ldsdma.h): no compiler-generated instruction may read or write it
    for o in outs:
        inasm = False
        for ln in open(o):
            t = ln.strip()
            if t.startswith(";;#ASMSTART"):
                inasm = True
            elif t.startswith(";;#ASMEND"):
                inasm = False
            elif not inasm and t and not t.startswith((";", ".")) and re.search(r"\bm0\b", t.split(";")[0]):
                raise AssertionError(f"{os.path.basename(o)}: compiler-generated use of m0: {t}")
