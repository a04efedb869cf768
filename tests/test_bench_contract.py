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
                '"kind"', '"sample"'):
        assert key in src, key
    assert "'--gpus'" in src and "'--steps'" in src and "'--warmup'" in src
