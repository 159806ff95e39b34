"""A stand-in `rdkit` package for TESTS (never importable by the product: only tests put tests/fake_rdkit on sys.path).

RDKit (`rdkit==2024.3.3`, reference enviroment.yaml) is installed neither in the build container nor on the GPU test box, so
the RDKit-facing code of physdock_amd (physics.rdkit_get_next_step_pos / rdkit_ref_mol_poses, mmff.terms_from_rdkit,
chirality.ChiralityReference.from_rdkit; reference models/model.py:26-52,188-203 and redocking.py:231-238) would otherwise
never execute.  This package implements exactly the slice of RDKit's Python API those functions call, on a synthetic molecule
(`FakeMol`, built from physdock_amd.mmff.synthetic_terms): parameter getters return the synthetic tables term by term, the
force field evaluates oracle/mmff_oracle.py, `MMFFOptimizeMolecule` runs the oracle's BFGS, and every call is recorded in
`CALLS` so that tests can assert the call SEQUENCE of the reference.  It says nothing about RDKit's arithmetic."""
__version__ = "fake-0 (tests/fake_rdkit)"

CALLS = []


def record(name, **kw):
    CALLS.append((name, kw))
