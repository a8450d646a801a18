"""The only numeric output of the reference that its tree holds: the accuracy lines of README.md:226-236 (committed as
tests/golden/readme_cora_2s_log.json).  They pin two things the oracle and the log contract rely on:
  * accuracies are reported in PER CENT with six decimals (sci::accuracy is external, gcn.h:626-630 only prints it);
  * the set sizes: trainSetSize = (uint64)(n * train_ratio), valSetSize likewise, the test set is the rest (gcn.h:560-562,
    621-626) - every sampled value must be an integer count over exactly that denominator."""
import json
import os
from fractions import Fraction

import cognn_oracle as co

HERE = os.path.dirname(os.path.abspath(__file__))


def _is_count_over(pct, denom):
    k = round(pct / 100.0 * denom)
    return 0 <= k <= denom and abs(100.0 * k / denom - pct) < 5e-7          # six printed decimals


def test_readme_accuracy_lines_are_per_cent_counts_over_the_oracle_set_sizes():
    fx = json.load(open(os.path.join(HERE, "golden", "readme_cora_2s_log.json")))
    n = fx["rows_per_party"]
    train = int(n * fx["train_ratio"]); val = int(n * fx["val_ratio"]); test = n - train - val
    assert (train, val, test) == (108, 108, 326)
    for ep in fx["epochs"]:
        assert _is_count_over(ep["full"], n) and _is_count_over(ep["train"], train) and _is_count_over(ep["test"], test)
        # any other split of the 542 rows would not produce these values
        assert not _is_count_over(ep["train"], train + 1) and not _is_count_over(ep["test"], test + 1)
        # border subsets: a count over SOME subset of the respective set
        for key, cap in (("border_train", train), ("border_test", test)):
            f = Fraction(ep[key] / 100.0).limit_denominator(cap)
            assert abs(100.0 * f.numerator / f.denominator - ep[key]) < 5e-7 and f.denominator <= cap


def test_readme_iteration_lines_pin_the_schedule():
    """Twelve '::iteration took' lines for two accuracy blocks: an epoch is 6 GAS iterations (gcn.h:929-942), and the two
    iterations of each epoch that run no message passing - the apply-only ones, ss_...h:709,941 - are by far the shortest:
    positions 2 and 4, as the oracle's schedule has them."""
    fx = json.load(open(os.path.join(HERE, "golden", "readme_cora_2s_log.json")))
    secs = fx["iteration_seconds"]
    p = co.GnnParam(num_labels=7, input_dim=12, hidden_dim=4, num_samples=8)
    o = co.OracleEngine(2, [0, 1, 2, 3], [1, 2, 3, 0], [0, 1, 0, 1], [[0.0] * 12] * 8, [0] * 8, p, seed=1)
    ep = o.epoch_len()
    assert len(secs) == len(fx["epochs"]) * ep == 12
    f = o.fwd_layers()
    apply_only = [e for e in range(ep) if e != 0 and e % f == 0]
    assert apply_only == [2, 4]
    for e0 in range(0, len(secs), ep):
        order = sorted(range(ep), key=lambda e: secs[e0 + e])
        assert sorted(order[:2]) == apply_only
        assert max(secs[e0 + e] for e in apply_only) < 0.3 * min(secs[e0 + e] for e in range(ep) if e not in apply_only)


def test_oracle_metrics_use_the_same_set_sizes_and_log_lines_print_per_cent(capsys):
    """The oracle's metric definitions on a 542-row party reproduce that structure, and print_metrics writes the reference's
    lines in per cent."""
    import numpy as np
    from cognn_amd.engine import print_metrics
    V, k = 1084, 2
    src, dst = co.synth_graph(V, 2000, 5)
    feats, labels = co.synth_features(V, 12, 7, 6, density=0.2)
    p = co.GnnParam(num_labels=7, input_dim=12, hidden_dim=4, num_samples=V, learning_rate=0.5, train_ratio=0.2, val_ratio=0.2, test_ratio=0.6)
    o = co.OracleEngine(k, src, dst, [v % k for v in range(V)], feats, labels, p, seed=3)
    o.run(2)
    m = [x for x in o.metrics if x["party"] == 0][0]
    assert m["n"] == 542
    for key, denom in (("full", 542), ("train", 108), ("test", 326)):
        assert _is_count_over(100.0 * m[key], denom), (key, m[key])
    print_metrics(m)
    out = capsys.readouterr().out
    assert "full set accuracy = %f\n" % (100.0 * m["full"]) in out and "border test set accuracy = %f\n" % (100.0 * m["border_test"]) in out
