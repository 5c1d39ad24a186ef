"""The plug-in's hardware-queue policy (dcvc_amd/plugin/inference_extensions_cuda.py::_hw_queue_policy): one
hardware queue per stream-priority level is the measured configuration; the drop-in sets it when it still can and WARNS when the
HIP runtime is already up with another value - it must not run at half speed silently. Child processes: the policy acts at import."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = ("import warnings, os, sys\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "with warnings.catch_warnings(record=True) as w:\n"
        "    warnings.simplefilter('always')\n"
        "    import inference_extensions_cuda as plug\n"
        "print('POLICY', plug.hw_queue_policy)\n"
        "print('ENV', os.environ.get('GPU_MAX_HW_QUEUES'))\n"
        "print('WARNED', any('GPU_MAX_HW_QUEUES' in str(x.message) for x in w))\n") % (ROOT, os.path.join(ROOT, "dcvc_amd", "plugin"))


def _run(env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "WORLD_SIZE")}
    env.update(env_extra)
    r = subprocess.run([sys.executable, "-c", CODE], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return dict(line.split(" ", 1) for line in r.stdout.strip().splitlines() if " " in line)


def test_the_plug_in_sets_one_queue_per_level_before_the_runtime_starts():
    out = _run({})
    assert out["ENV"] == "1" and "set by the plug-in" in out["POLICY"] and out["WARNED"] == "False"


def test_a_host_setting_is_respected_and_another_value_warns():
    out = _run({"GPU_MAX_HW_QUEUES": "1"})
    assert out["ENV"] == "1" and "set by the host" in out["POLICY"] and out["WARNED"] == "False"
    out = _run({"GPU_MAX_HW_QUEUES": "4"})
    assert out["ENV"] == "4" and "NOT the measured setting" in out["POLICY"] and out["WARNED"] == "True"


def test_a_job_of_several_ranks_is_left_alone():
    out = _run({"WORLD_SIZE": "2"})
    assert out["ENV"] == "None" and "several ranks" in out["POLICY"] and out["WARNED"] == "False"
