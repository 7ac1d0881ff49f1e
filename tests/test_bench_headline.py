"""bench.py prints its headline line even when an optional section (`also`, `strong`: the only code that talks point-to-point
over multi-GPU RCCL) hangs or raises: the watchdog of bench.Headline (exit code 4: the headline is printed, the run is not a success).  CPU-only: the class is driven directly in a child
process (it leaves through os._exit)."""
import json
import os
import subprocess
import sys

from conftest import ROOT

_SCRIPT = r"""
import sys, time
sys.path.insert(0, %(root)r)
import bench
bench.DETAIL_OUT = ""          # no bench_detail.json from a unit test
mode = sys.argv[1]
hl = bench.Headline(int(sys.argv[2]), {"metric": "m", "value": 1.0} if sys.argv[2] == "0" else None, 0.5)
hl.parity = True
if mode == "hang":
    hl.sections["also"] = {"verify": {"value": 2.0}}     # a section that finished before the hang is kept
    time.sleep(60)                                       # a rank stuck in a collective
    print("not reached")
elif mode == "raise":
    try:
        raise RuntimeError("NCCL error: unhandled system error")
    except Exception as e:
        hl.abandon("optional section failed on rank 0: %%s" %% e)
    print("not reached")
elif mode == "bad-parity":
    hl.parity = False
    time.sleep(60)
else:
    hl.sections["also"] = {"verify": {"value": 2.0}}
    hl.sections["strong"] = {}
    hl.done.set()
    hl.emit()
    hl.emit()                                            # the line is printed once
    time.sleep(1.0)                                      # past the deadline: the watchdog has gone away
    print("end")
"""


def _run(mode, rank="0"):
    r = subprocess.run([sys.executable, "-c", _SCRIPT % {"root": ROOT}, mode, rank], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=120)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, [json.loads(l) for l in lines]


def test_hang_in_an_optional_section_still_prints_the_headline():
    r, js = _run("hang")
    assert r.returncode == 4 and len(js) == 1 and "not reached" not in r.stdout
    j = js[0]
    assert j["value"] == 1.0 and j["parity_sample_ok"] is True and j["also"]["verify"]["value"] == 2.0
    assert "abandoned" in j["optional_sections"] and "abandoned" in r.stderr
    assert "bench_detail: {" in r.stderr                  # the full record goes to stderr, the compact one to stdout


def test_other_ranks_leave_silently():
    r, js = _run("hang", "1")
    assert r.returncode == 4 and js == [] and "not reached" not in r.stdout


def test_exception_in_an_optional_section_still_prints_the_headline():
    r, js = _run("raise")
    assert r.returncode == 4 and len(js) == 1 and "NCCL error" in js[0]["optional_sections"] and "not reached" not in r.stdout


def test_failed_parity_never_publishes_a_value():
    r, js = _run("bad-parity")
    assert r.returncode == 3 and js[0]["value"] is None and js[0]["parity_sample_ok"] is False


def test_normal_end_prints_once_without_a_note():
    r, js = _run("ok")
    assert r.returncode == 0 and len(js) == 1 and "optional_sections" not in js[0] and "strong" not in js[0]
    assert r.stdout.strip().endswith("end")
