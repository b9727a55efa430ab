"""The stdout line of bench.py must stay parseable by the driver: ONE JSON object below bench.LINE_LIMIT bytes carrying the contract's
headline fields + `config` + `roofline` (with `traffic`) + `cpu_baseline`; the full result tree goes to
bench_full.json / stderr.  Round 5's 20.5 KB line was not parsed -- these tests build the line from that very tree."""
import io
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
            'data', 'config', 'roofline', 'cpu_baseline')


def canned():
    return json.load(open(os.path.join(ROOT, 'profiles', 'r5_bench.json')))


def test_line_from_the_round5_tree_is_small_and_complete():
    full = canned()
    assert len(json.dumps(full)) > 3 * bench.LINE_LIMIT          # the tree that broke the parser
    line = bench.compact_line(full)
    txt = json.dumps(line)
    assert len(txt) < bench.LINE_LIMIT
    back = json.loads(txt)
    for k in CONTRACT:
        assert k in back, k
    assert back['value'] == pytest.approx(full['value'], rel=1e-6)
    assert back['ms_per_step'] == pytest.approx(full['ms_per_step'], rel=1e-6)
    assert back['config']['workload'].startswith('sample_MolDiff_simple.yml')
    assert 'model' not in back['config']
    r = back['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in r, k
    assert r['frac'] == pytest.approx(r['achieved'] / r['peak'], rel=1e-4)
    c = back['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in c, k
    # the other configurations: one-line summaries only
    for name, summ in back['configs'].items():
        assert len(json.dumps(summ)) < 300, name
        assert 'ms_per_step' in summ and 'value' in summ
    assert set(back['configs']) == set(full['configs'])
    assert 'speedup_vs_cpu_baseline' not in back       # ratios against the CPU are context, not on the line


def test_line_stays_below_the_limit_when_the_tree_grows():
    full = canned()
    full['config']['workload'] = 'w' * 5000
    full['cpu_baseline']['sample'] = 's' * 9000
    full['roofline']['kernel_symbol'] = 'k' * 4000
    full['dtype'] = 'd' * 3000
    for i in range(60):
        full['configs']['extra_%d' % i] = dict(full['configs']['guided'])
    line = bench.compact_line(full)
    assert len(json.dumps(line)) < bench.LINE_LIMIT
    for k in CONTRACT:
        assert k in line, k
    assert 'configs' in line.get('dropped', [])


def test_line_without_the_single_gpu_extras():
    full = {k: v for k, v in canned().items() if k not in ('configs', 'cpu_baseline', 'sample_wall_s', 'segment_sum', 'aggregation_large')}
    full['n_gpus'] = 8
    line = bench.compact_line(full)
    assert 'cpu_baseline' not in line and 'configs' not in line       # rank 0 at N = 1 only
    assert line['roofline']['traffic'] == pytest.approx(full['roofline']['traffic'], rel=1e-5)
    assert len(json.dumps(line)) < 3000


def test_emit_result_writes_one_parseable_last_line(tmp_path, monkeypatch, capfd):
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    monkeypatch.setattr(bench, '_REAL_STDOUT', None)
    bench.emit_result(canned())
    out, err = capfd.readouterr()
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1
    last = json.loads(lines[-1])
    assert len(lines[-1]) < bench.LINE_LIMIT and last['full'] == 'bench_full.json'
    tree = json.load(open(tmp_path / 'bench_full.json'))
    assert tree['configs']['guided']['roofline']['frac'] == canned()['configs']['guided']['roofline']['frac']
    assert any(ln.startswith('BENCH_FULL ') for ln in err.splitlines())


def test_round6_tree_keeps_the_training_probe_on_the_line():
    """The round-6 tree (profiles/r6_bench_full_driver.json: the driver's command) carries config #5's fixed-probe loss pair; the line keeps it
    as `configs.train.loss_fixed_probe` (two numbers, or a short message if the probe failed) and stays below the limit."""
    full = json.load(open(os.path.join(ROOT, 'profiles', 'r6_bench_full_driver.json')))
    line = bench.compact_line(full)
    assert len(json.dumps(line)) < bench.LINE_LIMIT
    tr = line['configs']['train']
    assert tr['ms_per_step'] == pytest.approx(full['configs']['train']['ms_per_step'], rel=1e-5)
    before, after = tr['loss_fixed_probe']
    assert after < before
    full['configs']['train']['loss_fixed_probe_before_after'] = ['RuntimeError(' + 'x' * 500 + ')'] * 2      # a failed probe: clipped text
    line = bench.compact_line(full)
    assert len(json.dumps(line)) < bench.LINE_LIMIT and all(len(x) <= 40 for x in line['configs']['train']['loss_fixed_probe'])
