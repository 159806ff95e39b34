"""PDB text of predicted poses, formatted on the device - SURVEY 8(f) row 3, the step after the sampler.

`FeatureLoader.write_pdb_block(x_pred, infer_meta_data, receptor_only, ligand_only)` (reference
PhysDock/data/feature_loader.py:1230-1283; its caller writes one file per pose, redocking.py:341-356) builds every ATOM /
HETATM record with a Python f-string per atom: ~2 k atoms x 64 poses per round.  Only 24 of the 80 columns of a record - the
three coordinates - depend on the pose.  `PdbTemplate` builds the other 56 columns ONCE per system on the host (the same
f-string, so names / residue ids / elements are the reference's byte for byte) and `pd_pdb_format` fills the coordinates for
all poses of a batch in one launch (one thread per output byte; HBM-bound: B x N x 81 bytes written).  The text returned is
identical to the reference's, character for character, for every coordinate that fits the 8.3 field; the reference lets
wider values push the rest of the line to the right (a malformed record), here they raise.
"""
from __future__ import annotations

from typing import List, Sequence

import torch

from . import ops

PDB_CHAIN_IDS = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789"       # feature_loader.py:20

#: element symbols by atomic number - 1, the index convention of the loader's `ref_element` (feature_loader.py:1244)
ELEMENTS = (
    "H He Li Be B C N O F Ne Na Mg Al Si P S Cl Ar K Ca Sc Ti V Cr Mn Fe Co Ni Cu Zn Ga Ge As Se Br Kr Rb Sr Y Zr Nb Mo Tc Ru Rh "
    "Pd Ag Cd In Sn Sb Te I Xe Cs Ba La Ce Pr Nd Pm Sm Eu Gd Tb Dy Ho Er Tm Yb Lu Hf Ta W Re Os Ir Pt Au Hg Tl Pb Bi Po At Rn Fr "
    "Ra Ac Th Pa U Np Pu Am Cm Bk Cf Es Fm Md No Lr Rf Db Sg Bh Hs Mt Ds Rg Cn Nh Fl Mc Lv Ts Og").split()

HEADER, FOOTER = "MODEL     1\n", "\nTER\nENDMDL\nEND"                               # feature_loader.py:1282


class PdbTemplate:
    """Pose-independent part of `write_pdb_block` for one system: the selected records' 80 columns with blank coordinates
    (`rows`, uint8 [N,81] incl. the newline) and the atom index of every record (`atom`, int32 [N])."""

    def __init__(self, infer_meta_data, receptor_only: bool = False, ligand_only: bool = False, device=None):
        if receptor_only and ligand_only:
            raise NotImplementedError()                                              # as the reference (:1278-1279)
        ccds = infer_meta_data["ccds"]
        inner = infer_meta_data["atom_id_to_conformer_atom_id"]
        chunk_sizes = [int(c) for c in infer_meta_data["conformer_id_to_chunk_sizes"]]
        chain_class = infer_meta_data["CHAIN_CLASS"]
        conf = infer_meta_data["CONF_META_DATA"]
        residue_index = [int(r) for r in infer_meta_data["residue_index"]]
        asym_id = [int(a) for a in infer_meta_data["asym_id"]]
        n_atoms = len(inner)
        lines: List[str] = []
        atoms: List[int] = []
        atom_offset = 0
        for ccd_id, (ccd, chunk, res_id) in enumerate(zip(ccds, chunk_sizes, residue_index)):
            idx = [int(i) for i in inner[atom_offset:atom_offset + chunk]]
            names = [conf[ccd]["ref_atom_name_chars"][i] for i in idx]
            elements = [ELEMENTS[int(conf[ccd]["ref_element"][i])] for i in idx]
            chain_tag = PDB_CHAIN_IDS[asym_id[ccd_id]]
            record = "HETATM" if chain_class[ccd_id] == "ligand" else "ATOM"
            keep = (record == "ATOM") if receptor_only else (record == "HETATM") if ligand_only else True
            for k, atom_name in enumerate(names):
                name = atom_name if len(atom_name) == 4 else f" {atom_name}"
                line = (f"{record:<6}{atom_offset + 1:>5} {name:<4}{'':>1}{ccd.split()[0][-3:]:>3} {chain_tag:>1}"
                        f"{res_id + 1:>4}{'':>1}   {'':>24}{1.00:>6.2f}{70.:>6.2f}          {elements[k]:>2}{0:>2}")
                if len(line) != 80:
                    raise ValueError(f"record {atom_offset + 1} does not fit the fixed-width PDB columns: {line!r}")
                if keep:
                    lines.append(line)
                    atoms.append(atom_offset)
                atom_offset += 1
                if atom_offset == n_atoms:
                    break
        self.n_atoms = n_atoms
        self.n_records = len(lines)
        raw = ("\n".join(lines) + "\n").encode("ascii") if lines else b""
        self.rows = torch.frombuffer(bytearray(raw), dtype=torch.uint8).reshape(self.n_records, 81) if lines else \
            torch.zeros(0, 81, dtype=torch.uint8)
        self.atom = torch.tensor(atoms, dtype=torch.int32)
        if device is not None:
            self.to(device)

    def to(self, device):
        self.rows, self.atom = self.rows.to(device), self.atom.to(device)
        return self

    def format(self, x_pred: torch.Tensor) -> torch.Tensor:
        """x_pred [B,A,3] (or [A,3]) on the device -> uint8 [B, N, 81]: the records of every pose, still on the device"""
        if x_pred.dim() == 2:
            x_pred = x_pred[None]
        if not x_pred.is_cuda:
            raise RuntimeError("PdbTemplate.format runs on an MI355X (HIP) device only; there is no CPU path")
        if self.rows.device != x_pred.device:
            self.to(x_pred.device)
        B, A = int(x_pred.shape[0]), int(x_pred.shape[1])
        if A < self.n_atoms:
            raise ValueError(f"x_pred has {A} atoms, the system has {self.n_atoms}")
        x = x_pred.float().contiguous()
        out = torch.empty(B, self.n_records, 81, dtype=torch.uint8, device=x.device)
        overflow = torch.zeros(1, dtype=torch.int32, device=x.device)
        ops.check(ops._lib.init().pd_pdb_format(ops.ptr(x), ops.ptr(self.rows), ops.ptr(self.atom), ops.ptr(out), ops.ptr(overflow),
                                                B, A, self.n_records, ops.stream()), "pd_pdb_format")
        if int(overflow.item()):
            raise ValueError(f"{int(overflow.item())} coordinates are not finite or do not fit the %8.3f PDB field")
        return out

    def blocks(self, x_pred: torch.Tensor) -> List[str]:
        """the `write_pdb_block` text of every pose of the batch (one device launch, one copy back)"""
        body = self.format(x_pred).cpu().numpy()
        return [HEADER + bytes(b).decode("ascii")[:-1] + FOOTER if self.n_records else HEADER + FOOTER for b in body]


def write_pdb_block(x_pred: torch.Tensor, infer_meta_data, receptor_only: bool = False, ligand_only: bool = False) -> str:
    """drop-in for `FeatureLoader.write_pdb_block` (feature_loader.py:1230-1283) for one pose x_pred [A,3] on the device"""
    return PdbTemplate(infer_meta_data, receptor_only, ligand_only).blocks(x_pred)[0]


def write_pdb_blocks(x_pred: torch.Tensor, infer_meta_data, receptor_only: bool = False, ligand_only: bool = False) -> Sequence[str]:
    """all poses [B,A,3] of a round at once (what redocking.py:341-356 loops over)"""
    return PdbTemplate(infer_meta_data, receptor_only, ligand_only).blocks(x_pred)
