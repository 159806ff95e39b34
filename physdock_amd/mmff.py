"""MMFF94 term tables for the on-device ligand relaxation (filled in below; see csrc/mmff.hip)."""
from __future__ import annotations


class MMFFTerms:       # placeholder, completed later in this round
    pass


def terms_from_rdkit(ref_mol, strict=False):
    if strict:
        raise NotImplementedError
    return None
