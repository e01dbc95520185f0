"""CPU: the measured table in DESIGN.md is the one tools/design_table.py generates from the committed bench line (profiles/r06_final_bench.json, the rocprofv3 trace
summary, profiles/traffic.json) -- a number quoted in the design document that no committed measurement backs would fail here --, and the trace summariser keeps launches of
different workloads apart even when they share a kernel and a grid (the balanced form is launched with one wavefront per wave slot whatever the batch)."""
import os
import subprocess
import sys

from conftest import ROOT


def test_design_table_matches_the_committed_bench_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "design_table.py"), "--check"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, "DESIGN.md's Measured (round 6) table differs from what tools/design_table.py generates from profiles/: run it\n" + r.stdout + r.stderr


def test_trace_summary_splits_one_grid_by_duration(tmp_path):
    trace = tmp_path / "t_kernel_trace.csv"
    head = "Kind,Agent_Id,Queue_Id,Stream_Id,Thread_Id,Dispatch_Id,Kernel_Id,Kernel_Name,Correlation_Id,Start_Timestamp,End_Timestamp,LDS_Block_Size,Scratch_Size,VGPR_Count,Accum_VGPR_Count,SGPR_Count,Workgroup_Size_X,Workgroup_Size_Y,Workgroup_Size_Z,Grid_Size_X,Grid_Size_Y,Grid_Size_Z\n"
    name = '"void ssim_hip::(anonymous namespace)::ssim_strip2_kernel<0, 0, true, true>(ssim_hip::(anonymous namespace)::KArgs)"'
    rows = []
    t = 0
    for i, dur in enumerate([2500000, 2600000, 90000, 95000, 88000, 2550000]):         # ns: three batch launches and three single-pair launches on the same grid
        rows.append("KERNEL_DISPATCH,1,1,1,1,%d,7,%s,%d,%d,%d,11264,0,116,0,96,64,1,1,131072,1,1\n" % (i, name, i, t, t + dur))
        t += dur + 1000
    trace.write_text(head + "".join(rows))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_prof.py"), "label", str(trace)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("| ssim_strip2_kernel<0, 0, true, true>")]
    assert len(lines) == 2, r.stdout
    cells = [[c.strip() for c in l.split("|")] for l in lines]
    assert [c[5] for c in cells] == ["3", "3"]                                          # launches per cluster
    assert abs(float(cells[0][6]) - 91.0) < 0.1 and abs(float(cells[1][6]) - 2550.0) < 0.1      # average us of each
