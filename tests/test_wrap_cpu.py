"""CPU: uint16 wrap-around of the coverage (SURVEY B-T1) - the flat restatement against the
reference's own Pile objects on the crafted inputs of tests/wrapcase.py."""
import numpy as np
import pytest

from oracle import oracle as ora
import wrapcase


def test_wrapped_values_survive_in_valid_regions():
    read_len, ov, kinds = wrapcase.wrap_inputs()
    o = ora.Oracle(read_len, ov, n_threads=4, ref=ora.have_ref())
    assert o.initialize() == 0
    p = o.piles()
    for t in list(kinds["fill"]) + list(kinds["dip"]) + list(kinds["edge"]):
        assert p["alive"][t], t
    wrapped = 0
    for t in kinds["fill"]:
        d = o.pile_data(t)
        big = np.nonzero(d >= 65000)[0]
        assert len(big) == 20 and big[-1] - big[0] == 19, (t, big)        # exactly [J-15, J+5)
        assert d[big[0]] == 65536 - (1 + t % 3)
        assert p["begin"][t] < big[0] and big[-1] < p["end"][t]           # inside the valid region
        wrapped += 1
    assert wrapped == len(kinds["fill"])
    for t in kinds["edge"]:
        d = o.pile_data(t)
        assert d.max() < 1000 and p["begin"][t] >= 300                    # wrapped stretch cut away


@pytest.mark.skipif(not ora.have_ref(), reason="reference objects not built here")
def test_flat_restatement_equals_reference_objects_on_wrap():
    read_len, ov, kinds = wrapcase.wrap_inputs(seed=1)
    res = []
    for ref in (False, True):
        o = ora.Oracle(read_len, ov, n_threads=4, ref=ref)
        assert o.construct() == 0
        o.remove_transitive_edges()
        p = o.piles()
        res.append((p, [o.pile_data(t) for t in range(24)], o.all_intervals(0), o.all_intervals(1), o.overlap_list(0),
                    o.edges()))
    a, b = res
    for k in a[0]:
        assert (a[0][k] == b[0][k]).all(), k
    for x, y in zip(a[1], b[1]):
        assert (x == y).all()
    for i in (2, 3):
        for x, y in zip(a[i], b[i]):
            assert (np.asarray(x) == np.asarray(y)).all()
    for i in (4, 5):
        for k in a[i]:
            assert (np.asarray(a[i][k]) == np.asarray(b[i][k])).all(), k
