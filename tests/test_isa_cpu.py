"""Build-time guards on the gfx950 code objects inside libmpv_hip.so (no GPU needed): register / LDS / scratch budgets of
the hot kernels -- the occupancy every measured number in DESIGN.md rests on -- and the CDNA4 instructions the design depends
on (16x16x32 bf16 MFMA, LDS-DMA, the transposing LDS read).  A compiler or source change that pushes the 256x256 GEMM past
256 VGPRs (spills) or the LayerNorm backward past 128 (3 instead of 4 waves per SIMD) fails here, on CPU."""
import os
import re
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "youku-mplug_amd", "libmpv_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"


def _code_objects():
    data = open(LIB, "rb").read()
    out = []
    for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data):
        p = m.start()
        (n,) = struct.unpack_from("<Q", data, p + 24)
        off = p + 32
        for _ in range(n):
            o, s, tl = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off:off + tl].decode()
            off += tl
            if "gfx950" in triple and s > 0:
                out.append(data[p + o:p + o + s])
    return out


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    if not os.path.isfile(os.path.join(LLVM, "llvm-readelf")):
        pytest.skip("ROCm LLVM tools not present")
    d = tmp_path_factory.mktemp("co")
    ks, files = {}, []
    for i, blob in enumerate(_code_objects()):
        f = str(d / f"co_{i}.elf")
        open(f, "wb").write(blob)
        files.append(f)
        txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f], capture_output=True, text=True).stdout
        for blk in txt.split("- .agpr_count:")[1:]:
            g = lambda key: int(re.search(r"\." + key + r":\s+(\d+)", blk).group(1))
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)          # mangled: template arguments read as ILb0ELi4EE etc.
            ks[name] = dict(vgpr=g("vgpr_count"), agpr=int(blk.split()[0]), sgpr=g("sgpr_count"), lds=g("group_segment_fixed_size"),
                           scratch=g("private_segment_fixed_size"), file=f)
    assert len(ks) > 80, f"only {len(ks)} kernels found in {LIB}"
    return ks, files


def _sel(ks, frag):
    r = {n: k for n, k in ks.items() if frag in n}
    assert r, frag
    return r


def test_gemm_kernels_fit_their_occupancy(kernels):
    ks, _ = kernels
    for n, k in _sel(ks, "14gemm256_kernelI").items():      # one 512-thread workgroup per CU = 2 waves per SIMD: 256 registers
        assert k["scratch"] == 0 and k["vgpr"] + k["agpr"] <= 256 and k["lds"] == 135168 + 1024, (n, k)      # ring / staged tile + the bias slice
    assert len(_sel(ks, "14gemm256_kernelI")) == 9          # fwd x {256,192,160}, dgrad x {256,192,160}, kmapped dgrad, wgrad, kmapped wgrad
    for n, k in _sel(ks, "16gemm_bf16_kernelI").items():    # two 256-thread workgroups per CU
        assert k["scratch"] == 0 and k["vgpr"] <= 256 and k["lds"] == 65536, (n, k)


def test_streaming_kernels_keep_their_waves(kernels):
    ks, _ = kernels
    budget = {"14ln_fwd8_kernelILi2EE": 72, "14ln_fwd8_kernelILi4EE": 128, "14ln_bwd8_kernelILi2ELb1ELb0EE": 128,      # <2>, <4>, <2, true>
              "14ln_bwd8_kernelILi2ELb1ELb1EE": 168,      # <2, true, PF>: two row sets, 3 waves per SIMD (768 workgroups)
              "20ln_bwd8_plain_kernelILi4ELb0EE": 168, "20ln_bwd8_plain_kernelILi4ELb1EE": 168, "20ln_stream_fwd_kernelILi4E": 168, "20temporal_attn_kernelILb0ELi8ELi96EE": 128, "20temporal_attn_kernelILb1ELi8ELi96EE": 128,
              # 4 frames (the shipped pre-train YAML) and 16 (the retrieval recipe; LDS allows 7 / 5 waves per CU there, registers are not the limit)
              "20temporal_attn_kernelILb0ELi4ELi96EE": 128, "20temporal_attn_kernelILb1ELi4ELi96EE": 128,
              "20temporal_attn_kernelILb0ELi16ELi96EE": 192, "20temporal_attn_kernelILb1ELi16ELi96EE": 192,
              # round 4: the bf16-row temporal attention (v_dot2c): same budgets as the fp32-row instances it replaces
              "24temporal_attn_b16_kernelILb0ELi8ELi96EE": 128, "24temporal_attn_b16_kernelILb1ELi8ELi96EE": 128,
              "24temporal_attn_b16_kernelILb0ELi4ELi96EE": 128, "24temporal_attn_b16_kernelILb1ELi4ELi96EE": 128,
              "24temporal_attn_b16_kernelILb0ELi16ELi96EE": 192, "24temporal_attn_b16_kernelILb1ELi16ELi96EE": 192,
              # the 16-wave weight-streaming instance: 1024 threads = 4 waves per SIMD
              "19gemm_small_m_kernelILb1ELi16ELi8EE": 128}
    for prefix, lim in budget.items():
        for n, k in _sel(ks, prefix).items():
            assert k["scratch"] == 0 and k["vgpr"] <= lim, (n, k, lim)


def test_attention_kernels_budgets(kernels):
    ks, _ = kernels
    for n, k in ks.items():
        if "attn_" not in n or "temporal" in n or "pair64" in n or "duo96" in n:      # the paired-block / duo kernels have their own tests below
            continue
        lean = "ELi320ELi3EE" in n                               # <..., 320, 3>: GPT instances, two 5-wave workgroups per CU, 3 waves per SIMD
        assert k["vgpr"] + (0 if "attn_bwd_dkv_kernel" in n else k["agpr"]) <= (170 if lean else 512), (n, k)
        # known small spill outside the tile loop of the 3-waves-per-SIMD dK/dV instance (48 bytes); anything larger is a regression
        assert k["scratch"] <= (48 if "attn_bwd_dkv_res_kernelILi64ELi320E" in n else 0), (n, k)


def test_paired_causal_attention_kernels_fit_four_items_per_cu(kernels):
    """csrc/attention_pair.inc: 3 waves and 40 KiB per (batch, head) item at S = 160 -> four items per CU only if a wave stays
    within the registers of 3 waves per SIMD (170) and nothing spills."""
    ks, _ = kernels
    pair = _sel(ks, "pair64_kernel")
    assert len(pair) == 6, sorted(pair)          # forward x {5, 7 blocks}, dQ x {3, 4 waves}, dK/dV x {3, 4 waves}
    for n, k in pair.items():
        limit = 256 if "ILi7ELi256ELi2EE" in n else 170      # the 7-block forward keeps 7 score tiles: two waves per SIMD
        spill = 80 if "dkv_pair64" in n else 0                 # the dK/dV role: 80 bytes outside the tile loop (as the one-shot instance)
        assert k["scratch"] <= spill and k["vgpr"] + k["agpr"] <= limit and k["lds"] == 0, (n, k)


def test_vit_duo_attention_kernels_fit_two_items_per_cu(kernels):
    """csrc/attention_duo.inc: 4 waves per (batch, head) item, two items per CU -> 2 waves per SIMD (256 registers), no scratch"""
    ks, _ = kernels
    duo = _sel(ks, "duo96_kernel")
    for n, k in duo.items():
        assert k["scratch"] == 0 and k["vgpr"] + k["agpr"] <= 256 and k["lds"] == 0, (n, k)


def test_persistent_attention_kernels_never_touch_scratch(kernels):
    """attn_*_pres_kernel: a scratch reload waits with s_waitcnt vmcnt(0), which would also drain the next item's LDS-DMA in the
    middle of the current item; every shipped instance must fit 256 registers (two waves per SIMD) without scratch, and the
    only vmcnt waits between the DMA issue and the end-of-item wait are the explicit ones (<= 2 per loop body: one per branch)."""
    ks, _ = kernels
    pres = {n: k for n, k in ks.items() if "_pres_kernel" in n}
    assert len(pres) == 2, sorted(pres)
    for n, k in pres.items():
        assert k["scratch"] == 0 and k["vgpr"] + k["agpr"] <= 256, (n, k)
    asm = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", next(iter(pres.values()))["file"]], capture_output=True, text=True).stdout
    for n in pres:
        body = asm[asm.index("<" + n + ">:"):]
        body = body[:body.index("s_endpgm")]
        first_loop_dma = [m.start() for m in re.finditer(r"buffer_load_dwordx4 v\d+, s\[\d+:\d+\], 0 offen lds", body)]
        assert len(first_loop_dma) >= 4, n                                  # prologue pair + loop pair
        loop = body[first_loop_dma[2]:]
        assert "scratch_" not in loop
        assert len(re.findall(r"s_waitcnt vmcnt\(", loop)) <= 2, (n, re.findall(r"s_waitcnt vmcnt\(\d+\)", loop))


def test_vit_attention_forward_instruction_budget(kernels):
    """attn_fwd_pres_kernel<96, 7, ..., LEAN> is bound by its instruction count (NOTEBOOK section 12 item 8): pin it.  Per work item and wave: 81
    MFMAs (7 x 6 score + 13 x 3 value products), exactly 100 exponentials (6 tiles x 16 + 4 of the 5-key tile), <= 100 packed bf16 conversions, and a
    static VALU count that must not creep (856 when round 6 moved `q * scale` into the qkv product's epilogue -- at run time that branch is
    skipped: 11.9 M VALU instructions per launch against 13.3 M, profiles/r06_final_sq_insts_attention_gemm.txt)."""
    ks, _ = kernels
    name = next(n for n in ks if "attn_fwd_pres_kernelILi96ELi7ELi8ELi512ELi1ELb1E" in n)
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--disassemble-symbols=" + name, ks[name]["file"]], capture_output=True, text=True).stdout
    body = dis[:dis.index("s_endpgm")]
    ops = re.findall(r"^\s+[0-9a-f]*\s*(v_\w+)", body, flags=re.M) or re.findall(r"\b(v_\w+)", body)
    mfma = sum(o.startswith("v_mfma") for o in ops)
    valu = len(ops) - mfma
    assert mfma == 81, mfma
    assert sum(o.startswith("v_exp_f32") for o in ops) == 100
    assert sum(o.startswith("v_cvt_pk_bf16_f32") for o in ops) <= 100
    assert valu <= 900, valu


def test_layernorm_backward_keeps_its_prefetch_in_flight(kernels):
    """ln_bwd8_kernel<2, true, PF>: the next row of a wave is requested before the current one is reduced.  That only works if the
    compiler waits for the CURRENT row with a counted s_waitcnt -- which it can only do when every request of a row is branch-free
    (buffer accesses; the statistics requested before the row's data): six 16-byte row loads per set, three sets in the code
    (prologue + the loop unrolled by two), and inside the loop no vmcnt wait below the six requests of the row in flight except the
    one at the loop head."""
    ks, _ = kernels
    name = next(n for n in ks if "14ln_bwd8_kernelILi2ELb1ELb1EE" in n)
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--disassemble-symbols=" + name, ks[name]["file"]], capture_output=True, text=True).stdout
    body = dis[:dis.index("s_endpgm")]
    loads = [m.start() for m in re.finditer(r"buffer_load_dwordx4 ", body)]
    assert len(loads) == 18, len(loads)
    assert "scratch_" not in body and "global_load_dwordx4" not in body[loads[0]:]
    loop = body[loads[5]:loads[17]]                                            # behind the prologue's requests .. last request of the third set
    waits = [int(w) for w in re.findall(r"s_waitcnt vmcnt\((\d+)\)", loop)]
    assert waits and sum(1 for w in waits if w < 6) <= 1, waits


def test_no_other_kernel_spills(kernels):
    ks, _ = kernels
    bad = {n: k["scratch"] for n, k in ks.items() if k["scratch"] and "attn_" not in n}
    assert not bad, bad


def test_gemm256_uses_the_cdna4_instructions_it_is_designed_on(kernels):
    ks, _ = kernels
    f = next(k["file"] for n, k in ks.items() if "14gemm256_kernelI" in n)
    asm = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", f], capture_output=True, text=True).stdout
    assert asm.count("v_mfma_f32_16x16x32_bf16") >= 9 * 100, "16x16x32 bf16 MFMA main loops"
    assert len(re.findall(r"buffer_load_dwordx4 .* lds", asm)) >= 9 * 8, "LDS-DMA ring fills"
    assert asm.count("ds_read_b64_tr_b16") >= 100, "transposing LDS reads of the dgrad / wgrad forms"
    assert "scratch_" not in asm


def test_temporal_attention_runs_on_the_packed_bf16_dot(kernels):
    """The bf16-row temporal attention exists because of v_dot2c_f32_bf16 (two MACs per instruction on bf16 pairs, fp32 accumulate):
    the instruction must be in the code objects, 6 * T per (T, forward) instance at head_dim 96 (12 chunks of 8 x 4 dot2 / 8) ..."""
    ks, _ = kernels
    f = next(k["file"] for n, k in ks.items() if "temporal_attn_b16_kernelILb0ELi8ELi96EE" in n)
    name = next(n for n in ks if "temporal_attn_b16_kernelILb0ELi8ELi96EE" in n)
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--disassemble-symbols=" + name, f], capture_output=True, text=True).stdout
    assert dis.count("v_dot2c_f32_bf16") + dis.count("v_dot2_f32_bf16") >= 48, "the bf16 dot product instruction is gone"


def _mem_sequence(dis):
    """memory-side skeleton of a disassembled kernel: L = global / buffer load, D = LDS-DMA, S = store, w<n> = s_waitcnt vmcnt(n), | = barrier"""
    seq = []
    for line in dis.split("\n"):
        t = line.split("//")[0].strip()
        m = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", t)
        if m:
            seq.append("w" + m.group(1))
        elif re.search(r"\b(global|buffer)_load\w* .*\blds\b", t):
            seq.append("D")
        elif re.search(r"\b(global|buffer)_load", t):
            seq.append("L")
        elif re.search(r"\b(global|buffer)_store", t):
            seq.append("S")
        elif "s_barrier" in t:
            seq.append("|")
    return seq


def test_requests_of_a_block_go_out_together(kernels):
    """Round 4's ISA audit, pinned: (i) a dQ block of the resident / paired / two-item attention backward asks for Q, dO, O and its
    statistic in ONE batch (>= 13 plain loads between the image DMAs and the first wait; they were four round trips behind each
    other while the row fragments were loaded inside `if (row < nrows)`); (ii) the 256x256 GEMM fetches no bias from global memory in
    its epilogue (the slice arrives by LDS-DMA in the prologue: no 8-byte global loads at all in the forward instance); (iii) the
    decoder's stream LayerNorm has no store pending at a barrier (no vmcnt(0) directly in front of a barrier behind a store)."""
    ks, _ = kernels

    def dis_of(frag):
        name = next(n for n in ks if frag in n)
        return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--disassemble-symbols=" + name, ks[name]["file"]], capture_output=True, text=True).stdout

    for frag, need in (("22attn_bwd_dq_res_kernelILi64ELi0ELi320ELi3EE", 13), ("25attn_bwd_dq_pair64_kernelILi192ELi3EE", 13), ("24attn_bwd_dq_duo96_kernelILi256EE", 19)):
        seq = _mem_sequence(dis_of(frag))
        first_dma = seq.index("D")
        run = 0
        for t in seq[first_dma:]:
            if t == "L":
                run += 1
            elif t.startswith("w") and run:
                break
        assert run >= need, (frag, run, seq[:40])
    assert "global_load_dwordx2" not in dis_of("14gemm256_kernelILb0ELb0ELb0ELi4EE")
    seq = _mem_sequence(dis_of("23ln_stream_fwd_wg_kernelILb0EE"))
    for i in range(1, len(seq) - 1):
        if seq[i] == "w0" and seq[i + 1] == "|":
            assert "S" not in seq[:i], seq
