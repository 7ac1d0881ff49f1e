"""tests/golden/gpu_expected.json is what the `-m gpu` tests compare against on the GPU box instead of running the pure-Python
oracle there.  Here, on the CPU: (i) the file is exactly what its committed generator produces from the
oracle, (ii) no GPU test takes the `pyoracle` fixture or imports the Python oracle."""
import ast
import glob
import json
import os
import subprocess
import sys

from conftest import ROOT, ints


def test_fixture_is_reproducible_from_the_python_oracle(tmp_path, pyoracle, golden):
    g = golden["gpu_expected"]
    o = pyoracle
    assert [ints(t) for t in g["torsion_points"]] == [o.mul_scalar(o.T8, j) for j in range(8)]
    assert ints(g["wide_bigint"]["result"]) == o.mul_scalar(o.B8, ints(g["wide_bigint"]["n"]))
    for c in g["sign_with_scalars"]:
        A, R, S = o.sign_with_scalars(ints(c["k"]), ints(c["rho"]), ints(c["msg"]))
        assert (A, R, S) == (ints(c["A"]), ints(c["R"]), ints(c["S"]))
        assert o.verify(A, R, S, ints(c["msg"])) is True and o.verify(A, R, S, ints(c["msg"]) + 1) is False
    for c in g["sign_schnorr_api"]:
        r, s = o.sign_schnorr_with_nonce(bytes.fromhex(c["key"]), ints(c["msg"]), ints(c["nonce"]))
        assert (tuple(r), s) == (ints(c["r"]), ints(c["s"]))
    # byte for byte: the committed generator writes the committed file
    gen = os.path.join(ROOT, "tests", "golden", "make_gpu_expected.py")
    src = open(gen).read().replace('os.path.join(HERE, "gpu_expected.json")', repr(str(tmp_path / "out.json")))
    script = tmp_path / "gen.py"
    script.write_text(src.replace("HERE = os.path.dirname(os.path.abspath(__file__))", "HERE = %r" % os.path.dirname(gen)))
    subprocess.run([sys.executable, str(script)], check=True, stdout=subprocess.PIPE)
    assert json.load(open(tmp_path / "out.json")) == g


def _is_gpu_marked(node, module_marked):
    if module_marked:
        return True
    for d in getattr(node, "decorator_list", []):
        if "mark.gpu" in ast.unparse(d):
            return True
    return False


def test_no_gpu_test_runs_the_python_oracle():
    offenders = []
    for path in sorted(glob.glob(os.path.join(ROOT, "tests", "test_*.py"))):
        src = open(path).read()
        tree = ast.parse(src)
        module_marked = any(isinstance(n, ast.Assign) and any(getattr(t, "id", "") == "pytestmark" for t in n.targets)
                            and "gpu" in ast.unparse(n.value) for n in tree.body)
        for node in tree.body:
            if isinstance(node, ast.FunctionDef) and node.name.startswith("test_") and _is_gpu_marked(node, module_marked):
                args = [a.arg for a in node.args.args]
                body = ast.unparse(node)
                if "pyoracle" in args or "bjj_oracle" in body:
                    offenders.append("%s::%s" % (os.path.basename(path), node.name))
    assert offenders == [], offenders
    # ... and the product, the bench and the smoke entry never import it either
    for rel in ("bench.py", "__graft_entry__.py", "babyjubjub-rs_amd/api.py", "babyjubjub-rs_amd/workload.py", "babyjubjub-rs_amd/shard.py"):
        assert "bjj_oracle" not in open(os.path.join(ROOT, rel)).read(), rel
