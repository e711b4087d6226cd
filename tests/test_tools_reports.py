"""The report tools that turn rocprofv3 output into the committed tables (tools/stream_kernel_tbps.py, tools/stream_gap_report.py): parsed on small
synthetic inputs, so that a change of the kernel's template signature or of the CSV columns is noticed on the CPU box."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tool, *args):
    return subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), *args], capture_output=True, text=True, check=True).stdout


def test_stream_kernel_tbps_reads_a_kernel_table(tmp_path):
    table = tmp_path / "stats.md"
    table.write_text("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---:|---:|---:|---:|---:|---:|\n"
                     "| `gemm_smallm_bf16_kernel<3, 16, 1, true, false, 1, false, true, false, true, false, false>` | 2400 | 15.0 | 6.25 | 5.6 | 9.1 | 16.7 |\n"
                     "| `gemm_smallm_bf16_kernel<2, 8, 1, false, true, 2, false, true, false, false, true, true>` | 920 | 8.6 | 9.40 | 8.8 | 11.6 | 9.4 |\n"
                     "| `stream_attention_tiles_kernel<32>` | 1200 | 9.6 | 8.0 | 5.7 | 11.2 | 10.4 |\n")
    out = _run("stream_kernel_tbps.py", str(table), "bf16")
    assert "ffn fc2 + residual (K 4096)" in out and "| 8.39 | 1.34 |" in out                 # 1024 x 4096 bf16 = 8.39 MB in 6.25 us
    assert "the previous block's final norm" in out and "stream_attention" not in out
    assert "all product launches together" in out


def test_stream_kernel_tbps_on_the_committed_tables():
    for mode in ("bf16", "fp32"):
        src = os.path.join(ROOT, "profiles", f"r06_m2_stream_{mode}_kernel_stats.md")
        want = open(os.path.join(ROOT, "profiles", f"r06_m2_stream_{mode}_kernel_tbps.md")).read()
        assert os.path.exists(src)
        got = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stream_kernel_tbps.py"), f"profiles/r06_m2_stream_{mode}_kernel_stats.md", mode],
                             capture_output=True, text=True, check=True, cwd=ROOT).stdout          # (the tool prints the path it was given: run from the root)
        assert got == want, f"profiles/r06_m2_stream_{mode}_kernel_tbps.md is not what the tool writes from the committed kernel table"


def test_stream_gap_report_reads_a_kernel_trace(tmp_path):
    rows = ["Kind,Agent_Id,Queue_Id,Kernel_Id,Kernel_Name,Start_Timestamp,End_Timestamp"]
    t = 1000
    for chunk in range(8):
        names = ["mel_logmel_kernel<true, 4>"] + ["void pk::gemm_smallm_bf16_kernel<3>(pk::GemmArgs)"] * 160
        for i, n in enumerate(names):
            gap = 8000 if i == 50 else 100                                                  # one 8 us hole per chunk, 0.1 us elsewhere
            t += gap
            rows.append(f'KERNEL_DISPATCH,1,1,1,"{n}",{t},{t + 5000}')
            t += 5000
    f = tmp_path / "kt_kernel_trace.csv"
    f.write_text("\n".join(rows) + "\n")
    out = _run("stream_gap_report.py", str(f))
    assert "161 kernels each" in out and "gaps > 5 us: 1.0 per chunk, 8.0 us per chunk" in out
    assert "in front of" in out and "gemm_smallm_bf16_kernel" in out
