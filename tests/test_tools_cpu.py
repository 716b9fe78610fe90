"""CPU tests of the measurement tools that cannot be exercised on the 1-GPU boxes: the all-reduce / backward overlap analysis
(tools/rocpd_overlap.py) on a synthetic kernel trace, so that it works first time when an N >= 2 trace exists."""
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _trace():
    # (name, queue, stream, start ns, end ns): compute on queue 1 (and a second compute queue 3), RCCL on queue 2
    return [("gemm256_kernel<false,false,false,4>(GemmArgs)", 1, 1, 0, 1000),
            ("ln_bwd8_kernel<2,true>(LnBwdArgs)", 1, 1, 1000, 1500),
            ("gemm256_kernel<true,true,false,4>(GemmArgs)", 3, 3, 1200, 1800),            # overlaps the LN on the main queue
            ("ncclDevKernel_Generic(ncclDevKernelArgs)", 2, 2, 500, 1700),                 # 1200 ns: [500,1000) + [1000,1500) + [1200,1700) -> fully covered
            ("ncclDevKernel_Generic(ncclDevKernelArgs)", 2, 2, 2000, 3000),                # 1000 ns: compute only on [2500, 2750)
            ("adamw_grouped_kernel", 1, 1, 2500, 2750),
            ("rcclSomething", 2, 2, 4000, 4100)]                                           # nothing beside it


def test_overlap_report_on_synthetic_trace():
    import rocpd_overlap as ro
    assert ro.covered(0, 10, [(2, 4), (3, 6), (8, 20), (-5, 1)]) == 4 + 2 + 1
    lines, share = ro.overlap_report(_trace())
    assert "3 RCCL kernel launches on queues [2]" in lines[0] and "4 compute launches on queues [1, 3]" in lines[0]
    assert abs(share - (1200 + 250 + 0) / (1200 + 1000 + 100)) < 1e-9
    rows = [l for l in lines if l.startswith("| `")]
    assert "100 %" in rows[0] and "25 %" in rows[1] and " 0 %" in rows[2]
    _, none = ro.overlap_report([k for k in _trace() if "ccl" not in k[0].lower()])
    assert none is None


def test_overlap_tool_reads_a_rocpd_database(tmp_path):
    db = str(tmp_path / "t.db")
    c = sqlite3.connect(db)
    c.execute("create table kernels (name text, queue_id int, stream_id int, start int, end int)")
    c.executemany("insert into kernels values (?,?,?,?,?)", _trace())
    c.commit()
    c.close()
    out = str(tmp_path / "o.md")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocpd_overlap.py"), db, out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "under compute kernels of another queue: 63 %" in open(out).read()


def test_zimage_layout_simulation():
    """The permuted 128-byte-row LDS image of csrc/attention_pair.inc: every fragment address returns its logical element and the
    hardware lane groups are bank-conflict-free (host simulation, tools/probe/sim_zimage_layout.py)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("sim_zimage_layout", os.path.join(ROOT, "tools", "probe", "sim_zimage_layout.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.check(160, verbose=False) and mod.check(224, verbose=False) and mod.check(32, verbose=False)
    assert mod.check_rot192(224, verbose=False)      # the head_dim-96 images of csrc/attention_duo.inc


def test_rocpd_gaps_on_a_synthetic_trace(tmp_path, capsys):
    """tools/rocpd_gaps.py (idle time between dispatches, by the kernel that follows the gap) on a hand-built rocpd table: three
    steady-state steps after the first optimizer launch, a 3 us gap in front of every `b` kernel, one 400 us host pause that must not
    count as a launch gap."""
    import sqlite3
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import rocpd_gaps
    db = str(tmp_path / "t.db")
    c = sqlite3.connect(db)
    c.execute("create table kernels (name text, start integer, end integer)")
    t = 0
    rows = [("adamw_first", t, t + 1000)]
    t += 1000
    for step in range(3):
        for name, dur, gap in (("a", 10000, 0), ("b", 20000, 3000), ("a", 10000, 0), ("adamw_grouped_kernel", 5000, 0)):
            t += gap
            rows.append((name, t, t + dur))
            t += dur
        t += 400000 if step == 0 else 0
    c.executemany("insert into kernels values (?, ?, ?)", rows)
    c.commit()
    c.close()
    sys.argv = ["rocpd_gaps.py", db, str(tmp_path / "o.md")]
    rocpd_gaps.main()
    out = open(tmp_path / "o.md").read()
    assert "12 dispatches over 3 steady-state steps" in out
    assert "idle between dispatches 0.00 ms/step" in out or "idle between dispatches 0.003 ms/step" in out or "0.00 ms/step" in out
    assert "| `b` | 3 | 0.01 | 3.00 |" in out, out
    assert "pauses > 200 us (host bookkeeping between steps): 0.4 ms" in out, out


def test_g256p_probe_layout_model():
    """tools/probe/sim_g256p_layout.py: the index arithmetic of the persistent 4-wave GEMM probe (tools/probe/g256p_probe.hip, not
    a product path, not yet run on a GPU) -- DMA image / fragment reads / bank conflicts / the permlane way out -- holds on the host."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("sim_g256p_layout", os.path.join(ROOT, "tools", "probe", "sim_g256p_layout.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    assert m.check_fragments() == 1          # conflict-free under the hardware lane groups of ds_read_b128
    assert m.check_way_out(768) and m.check_way_out(2304)
    spec = importlib.util.spec_from_file_location("sim_g256p_ring", os.path.join(ROOT, "tools", "probe", "sim_g256p_ring.py"))
    r = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(r)
    for tiles in (1, 2, 4):                  # the ring protocol: no refill without a barrier behind the last read, no read before a counted wait + barrier
        for nk in (10, 12, 37):
            assert r.run(tiles, nk) > 0


def test_bench_clock_power_sampler_without_and_with_a_source():
    """bench.py's ClockPowerSampler (shader clock / package power of the timed region in the JSON line): on a host without an SMI
    source it reports nulls and starts no thread; with a reader it samples at its period from a host thread and reports medians,
    the clock minimum and the power maximum."""
    import importlib.util
    import time
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    s = bench.ClockPowerSampler(0)
    if s.source is None:                       # (this container: no GPU, no SMI)
        with s:
            time.sleep(0.05)
        r = s.summary()
        assert r["sclk_mhz"] is None and r["power_w"] is None and r["clock_power_samples"] == 0 and r["clock_power_source"] is None
    seq = iter([(2100.0, 1300.0), (2050.0, 1350.0), (None, 1200.0), (2080.0, None)] + [(2070.0, 1310.0)] * 1000)
    s = bench.ClockPowerSampler(0, hz=200.0)
    s._read, s.source = (lambda: next(seq)), "fake"
    with s:
        t0 = time.time()
        while len(s.samples) < 8 and time.time() - t0 < 5.0:      # (a loaded host may schedule the sampling thread late)
            time.sleep(0.02)
    r = s.summary()
    assert r["clock_power_source"] == "fake" and r["clock_power_samples"] >= 5
    assert r["sclk_mhz_min"] == 2050.0 and r["power_w_max"] == 1350.0 and r["sclk_mhz"] == 2070.0 and r["power_w"] == 1310.0


def test_floor_table_tool_on_the_committed_profiles():
    """tools/floor_table.py: on the committed in-step GEMM table + kernel trace of the closing set it reproduces the
    committed floor table -- GEMM floor = executed FLOPs / 1.7 PF (or bytes / 6.3 TB/s where larger), every non-GEMM family priced on its
    algorithmic bytes, the step floor below north_star's 61 ms."""
    import re
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join("tools", "floor_table.py"), os.path.join("profiles", "r06_final_gemm_in_step_by_shape.md"),
                          os.path.join("profiles", "r06_final_kernel_trace.md"), "7"], capture_output=True, text=True, check=True, cwd=ROOT).stdout
    assert out == open(os.path.join(ROOT, "profiles", "r06_floor_table_final_tree.md")).read()
    m = re.search(r"Step floor of the present decomposition: ([\d.]+) \(GEMM\) \+ ([\d.]+) \(everything else\) = ([\d.]+) ms", out)
    g, o, t = (float(x) for x in m.groups())
    assert abs(g + o - t) < 0.11 and 30 < g < 40 and 10 < o < 15 and t < 61.0
    fam = re.search(r"\*\*GEMM family\*\* \| \| \| \| \| \| \*\*([\d.]+)\*\* \| \| \| \*\*([\d.]+)\*\*", out)
    assert float(fam.group(1)) > float(fam.group(2)) > 0

