"""wQMC line formatter (run_inference.py:254-305): hand-checked expectations, for the oracle's
restatement and for the product's native-backed functions of the same names."""
import pytest

from oracle import qmc_format as O
from tetrad_amd import qmc_format as P

MODS = [pytest.param(O, id="oracle"), pytest.param(P, id="native")]


ROWS = [
    # a b c d   s0        s1        s2      topo nsnps
    "0\t1\t2\t3\t1.000000\t4.000000\t3.000000\t0\t120\n",
    "0\t1\t2\t4\t5.000000\t2.000000\t6.000000\t1\t80\n",
    "0\t1\t3\t4\t9.000000\t8.000000\t0.500000\t2\t15\n",
    "0\t2\t3\t4\t0.001000\t0.001000\t0.001000\t0\t0\n",       # zero-data row: always dropped (min_snps >= 1)
]


@pytest.mark.parametrize("M", MODS)
def test_weight_strategies(tmp_path, M):
    iter_qmc_formatted = M.iter_qmc_formatted
    f = tmp_path / "q.tsv"
    f.write_text("".join(ROWS))
    # weights=0: unweighted, topology decides the split: 0 -> ab|cd, 1 -> ac|bd, 2 -> ad|bc
    assert list(iter_qmc_formatted(f, 0)) == ["0,1|2,3:1.00000", "0,2|1,4:1.00000", "0,4|1,3:1.00000"]
    # weights=1: mean of the two larger scores
    assert list(iter_qmc_formatted(f, 1)) == ["0,1|2,3:3.50000", "0,2|1,4:5.50000", "0,4|1,3:8.50000"]
    # weights=2: that mean / smallest score
    assert list(iter_qmc_formatted(f, 2)) == ["0,1|2,3:3.50000", "0,2|1,4:2.75000", "0,4|1,3:17.00000"]
    # weights=3: 1 - smallest / sum
    assert list(iter_qmc_formatted(f, 3)) == ["0,1|2,3:0.87500", "0,2|1,4:0.84615", "0,4|1,3:0.97143"]
    # filters
    assert list(iter_qmc_formatted(f, 1, min_snps=100)) == ["0,1|2,3:3.50000"]
    assert list(iter_qmc_formatted(f, 1, min_ratio=3.0)) == ["0,1|2,3:3.50000", "0,4|1,3:8.50000"]


@pytest.mark.parametrize("M", MODS)
def test_write_is_a_seeded_permutation(tmp_path, M):
    iter_qmc_formatted, write_qmc_format = M.iter_qmc_formatted, M.write_qmc_format
    f = tmp_path / "q.tsv"
    f.write_text("".join(ROWS * 50))
    write_qmc_format(f, tmp_path / "a.txt", weights=1, seed=1)
    write_qmc_format(f, tmp_path / "b.txt", weights=1, seed=1)
    a = (tmp_path / "a.txt").read_text()
    assert a == (tmp_path / "b.txt").read_text()
    assert sorted(a.splitlines()) == sorted(list(iter_qmc_formatted(f, 1)))
    assert a.splitlines() != list(iter_qmc_formatted(f, 1))
