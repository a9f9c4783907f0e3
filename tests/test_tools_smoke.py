"""Every script under tools/ that produces a file under profiles/ runs for one small iteration (`-m gpu`); the readers of rocprofv3 databases and
the shell drivers are checked for syntax (their inputs only exist under a profiler)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOLS = os.path.join(ROOT, "tools")

# script -> (argv, env): the smallest setting each script offers
RUNS = {
    "attn_l2_probe.py": ([], {"ROWS": "32"}),
    "attn_timeline.py": ([], {"ROWS": "32"}),
    "coresident_probe.py": (["1"], {"PAIRS": "10", "REP": "1"}),
    "flat_timeline.py": ([], {}),
    "fused_unit_timeline.py": ([], {"B": "32"}),
    "gemm_x3_timeline.py": (["1024", "512", "512"], {}),
    "hash_encoder.py": ([], {}),
    "hash_inference.py": ([], {}),
    "hash_train_step.py": ([], {}),
    "kernel_resources.py": ([os.path.join(ROOT, "lip2speech_amd", "csrc", "decoder_kernels.hip")], {}),
    "overlap_stamps.py": (["1"], {"PAIRS": "10", "S": "10", "CHAINS": "2"}),
    "pdecode_timeline.py": ([], {"B": "2", "S": "20"}),
    "pmc_dense.py": ([], {"ROWS": "32"}),
    "prof_decode.py": ([], {"ROWS": "32", "REPS": "1"}),
    "prof_train.py": ([], {}),
    "s2_unit_timeline.py": ([], {"B": "32"}),
    "skinny_timeline.py": ([], {"ROWS": "32"}),
    "steps20_timeline.py": ([], {"K": "3", "REP": "1"}),
    "time_evaluate_net.py": ([], {"N": "2", "ITERS": "4"}),
    "time_frontend.py": ([], {}),
    "time_group.py": ([], {"G": "1", "NT": "1"}),
    "time_latency.py": ([], {"ROWS": "2", "REPS": "1", "S": "20"}),
    "time_trunk.py": ([], {"B": "32"}),
    "time_step_phases.py": (["1"], {}),
    "time_vocoder.py": ([], {"N": "32", "ITERS": "4"}),
    "train_stages.py": ([], {}),
}
SYNTAX_ONLY = ["pmc_decode_json.py", "pmc_read.py", "rocprof_concurrency.py", "rocprof_summary.py"]


def test_tools_inventory_is_what_the_readme_lists():
    have = sorted(f for f in os.listdir(TOOLS) if f.endswith((".py", ".sh")))
    want = sorted(list(RUNS) + SYNTAX_ONLY + ["ab_bench.sh", "attn_l2_sweep.sh", "gpurun_retry.sh", "pmc_dense_kernels.sh", "pmc_step_kernels.sh", "profile_r5.sh", "profile_r5_chains.sh", "profile_r6.sh"])
    assert have == want, (set(have) ^ set(want))
    readme = open(os.path.join(TOOLS, "README.md")).read()
    for f in have + ["membw/membw.hip", "persist/persist_probe.hip"]:
        assert f in readme, f"tools/README.md does not mention {f}"
    for f in SYNTAX_ONLY:
        subprocess.run([sys.executable, "-m", "py_compile", os.path.join(TOOLS, f)], check=True)
    for f in have:
        if f.endswith(".sh"):
            subprocess.run(["bash", "-n", os.path.join(TOOLS, f)], check=True)
        else:
            assert "profiles/" in open(os.path.join(TOOLS, f)).read(), f"{f}: name the profiles/ file it produces in its docstring"


@pytest.mark.gpu
@pytest.mark.parametrize("script", sorted(RUNS))
def test_tool_runs_one_iteration(script):
    argv, env = RUNS[script]
    r = subprocess.run([sys.executable, os.path.join(TOOLS, script)] + argv, cwd=ROOT, env=dict(os.environ, **env), capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.strip(), "no output"
