"""rdkit.Chem.AllChem slice: MMFFOptimizeMolecule (oracle BFGS) and EmbedMultipleConfs (perturbed copies)"""
import numpy as np

from .. import record


def MMFFOptimizeMolecule(mol, mmffVariant="MMFF94", maxIters=200, nonBondedThresh=100.0, confId=-1,
                         ignoreInterfragInteractions=True):
    import mmff_oracle
    conf = mol.GetConformer(confId)
    record("MMFFOptimizeMolecule", mmffVariant=mmffVariant, maxIters=maxIters,
           ignoreInterfragInteractions=ignoreInterfragInteractions, start=conf.GetPositions())
    out = mmff_oracle.minimize(conf.GetPositions(), mol.terms.as_numpy(), max_iters=int(maxIters))
    pos = out[0] if isinstance(out, tuple) else out
    conf._pos[:] = np.asarray(pos, dtype=np.float64).reshape(-1, 3)
    return 1


def EmbedMultipleConfs(mol, numConfs=10, enforceChirality=True, **kw):
    record("EmbedMultipleConfs", numConfs=numConfs, enforceChirality=enforceChirality)
    base = mol.GetConformer().GetPositions()
    mol.RemoveAllConformers()
    rng = np.random.default_rng(numConfs)
    n_ok = max(1, numConfs - 2)                      # two embeddings "fail": the reference leaves their rows at zero
    return [mol.AddConformer(base + rng.normal(0, 0.05, size=base.shape)) for _ in range(n_ok)]
