"""Stereocentre perception from a bond graph (physdock_amd/chirality.py; reference redocking.py:231-238 lets RDKit's
FindMolChiralCenters decide).  Hand-built molecules whose answer is textbook chemistry; host logic only."""
from physdock_amd.chirality import centres_from_bonds, symmetry_classes

C, N, O, F, S, Cl, Br, H = 6, 7, 8, 9, 16, 17, 35, 1


def centre_atoms(n, bonds, el, **kw):
    return [c[0] for c in centres_from_bonds(n, bonds, elements=el, **kw)]


def test_real_stereocentre_with_explicit_hydrogen():
    # CHFClBr: C0 bonded to H1 F2 Cl3 Br4
    assert centre_atoms(5, [(0, 1), (0, 2), (0, 3), (0, 4)], [C, H, F, Cl, Br]) == [0]


def test_alanine_like_centre_with_implicit_hydrogen():
    # N0 - C1(-C2 methyl)(-C3(=O4)O5): heavy atoms only; C1 carries one implicit H
    bonds, orders = [(0, 1), (1, 2), (1, 3), (3, 4), (3, 5)], [1, 1, 1, 2, 1]
    el = [N, C, C, C, O, O]
    assert centre_atoms(6, bonds, el, bond_orders=orders) == []                       # four explicit neighbours required
    assert centre_atoms(6, bonds, el, bond_orders=orders, min_neighbours=3) == [1]    # implicit H on the sp3 carbon
    assert centre_atoms(6, bonds, el, bond_orders=orders, implicit_h=[2, 1, 3, 0, 0, 1]) == [1]


def test_gem_dimethyl_and_isopropyl_are_not_centres():
    # C0(-C1)(-C2)(-O3)(-N4): two methyls
    assert centre_atoms(5, [(0, 1), (0, 2), (0, 3), (0, 4)], [C, C, C, O, N]) == []
    # isopropyl CH: C0(-C1)(-C2)(-O3) + implicit H
    assert centre_atoms(4, [(0, 1), (0, 2), (0, 3)], [C, C, C, O], min_neighbours=3) == []


def test_cf3_tbutyl_sulfonyl_are_not_centres():
    # CF3 on a carbon chain: C0(F1)(F2)(F3)-C4-O5
    assert centre_atoms(6, [(0, 1), (0, 2), (0, 3), (0, 4), (4, 5)], [C, F, F, F, C, O]) == []
    # t-butyl: C0(C1)(C2)(C3)-O4
    assert centre_atoms(5, [(0, 1), (0, 2), (0, 3), (0, 4)], [C, C, C, C, O]) == []
    # sulfonyl: C0-S1(=O2)(=O3)-N4
    assert centre_atoms(5, [(0, 1), (1, 2), (1, 3), (1, 4)], [C, S, O, O, N], bond_orders=[1, 2, 2, 1]) == []


def test_substituents_that_differ_only_far_away_are_told_apart():
    # C0 with two propyl-like arms that differ at the third atom: C0(-C1-C2-O3)(-C4-C5-N6)(-F7)(-Cl8)
    bonds = [(0, 1), (1, 2), (2, 3), (0, 4), (4, 5), (5, 6), (0, 7), (0, 8)]
    el = [C, C, C, O, C, C, N, F, Cl]
    assert centre_atoms(9, bonds, el) == [0]
    # identical arms -> not a centre
    el2 = [C, C, C, O, C, C, O, F, Cl]
    assert centre_atoms(9, bonds, el2) == []


def test_ring_centre_and_symmetric_ring_atom():
    # methylcyclohexane-like ring C0..C5, methyl C6 on C0, OH O7 on C0: C0's two ring arms are equivalent -> no centre
    ring = [(i, (i + 1) % 6) for i in range(6)]
    assert centre_atoms(8, ring + [(0, 6), (0, 7)], [C] * 7 + [O]) == []
    # a second substituent on C2 breaks the ring symmetry: C0 and C2 become centres (cis / trans isomers exist)
    got = centre_atoms(10, ring + [(0, 6), (0, 7), (2, 8), (2, 9)], [C] * 7 + [O, F, Cl])
    assert got == [0, 2]


def test_three_coordinate_nitrogen_is_not_a_centre():
    assert centre_atoms(4, [(0, 1), (0, 2), (0, 3)], [N, C, O, F], min_neighbours=3) == []


def test_symmetry_classes_of_a_symmetric_molecule():
    # propane C0-C1-C2: the two ends share a class
    cls = symmetry_classes(3, [(0, 1), (1, 2)], [C, C, C])
    assert cls[0] == cls[2] != cls[1]


def test_without_elements_every_four_coordinate_atom_is_returned():
    # legacy behaviour: no chemistry, the caller's atoms keep their handedness
    assert centres_from_bonds(5, [(0, 1), (0, 2), (0, 3), (0, 4)]) == [(0, 1, 2, 3)]
