"""bench.py's JSON contract (static checks; the numbers themselves need the GPU)."""
import ast
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_has_contract_keys():
    src = open(os.path.join(ROOT, 'bench.py')).read()
    ast.parse(src)
    for key in ('"metric"', '"value"', '"unit"', '"n_gpus"', '"steps"', '"warmup"', '"ms_per_step"',
                '"higher_is_better"', '"scaling"', '"vs_baseline"', '"dtype"', '"data"', '"config"', '"workload"',
                '"roofline"', '"bound"', '"achieved"', '"peak"', '"frac"', '"traffic"', '"cpu_baseline"', '"cores"',
                '"kind"', '"sample"', '"members_total"', '"rccl_ranks"', '"per_rank"', '"kernel_ms"', "'value_weak'", "'secondary'"):
        assert key in src, key
    assert "'--gpus'" in src and "'--steps'" in src and "'--warmup'" in src


def test_default_scaling_is_baseline_c3_as_stated():
    """BASELINE C3 = ONE 100 000-member ensemble sharded over 1 -> 8 GPUs: `python bench.py --gpus N` must time that (strong),
    and never quietly something else; the other configs keep per-GPU sizes (weak)."""
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    bench = importlib.import_module('bench')
    a = bench.parse_args(['--gpus', '8'])
    assert a.scaling is None and a.config == 'c3' and bench.CONFIGS['c3']['scaling'] == 'strong' and bench.CONFIGS['c3']['members'] == 100000
    assert all(bench.CONFIGS[c]['scaling'] == 'weak' for c in ('c2', 'c4', 'c5'))
    assert bench.parse_args(['--scaling', 'weak']).scaling == 'weak'
    src = open(os.path.join(ROOT, 'bench.py')).read()
    assert 'raise SystemExit(3)' in src and 'table left in HBM on every rank' not in src       # no silent fallback of what `value` means
