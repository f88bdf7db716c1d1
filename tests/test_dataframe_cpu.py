"""``baybe_amd.dataframe.fuzzy_row_match`` against a brute-force restatement of the reference's match-matrix
algorithm (``baybe/utils/dataframe.py:361-460``): per parameter a ``len(right) x len(left)`` equality / nearest-value
matrix, AND-ed; first match per right row, unmatched rows dropped, a warning for multiple matches."""

import warnings

import numpy as np
import pandas as pd
import pytest

from baybe_amd.dataframe import FuzzyRowMatcher, SearchSpaceMatchWarning, fuzzy_row_match


class P:
    def __init__(self, name, numerical, discrete=True):
        self.name, self.is_numerical, self.is_discrete = name, numerical, discrete


def reference_match(left, right, parameters):
    cat = [p.name for p in parameters if not p.is_numerical and p.is_discrete]
    num = [p.name for p in parameters if p.is_numerical and p.is_discrete]
    m = np.ones((len(right), len(left)), dtype=bool)
    for c in cat:
        m &= np.asarray(right[c])[:, None] == np.asarray(left[c])[None, :]
    for c in num:
        ad = np.abs(np.asarray(right[c], dtype=float)[:, None] - np.asarray(left[c], dtype=float)[None, :])
        m &= ad == ad.min(axis=1, keepdims=True)
    first = left.index[m.argmax(axis=1)]
    return pd.Index(first[m.any(axis=1)]), right.index[m.sum(axis=1) > 1].tolist()


@pytest.mark.parametrize("seed", range(6))
def test_matches_the_reference_algorithm_on_random_spaces(seed):
    rng = np.random.default_rng(seed)
    n_left, n_right = 400, 60
    levels = {"a": np.array([0.0, 0.5, 1.0, 2.0, 4.0]), "b": np.arange(7) / 3.0, "c": np.array([10.0, 20.0])}
    left = pd.DataFrame({k: rng.choice(v, n_left) for k, v in levels.items()})
    left["cat"] = rng.choice(["x", "y", "z"], n_left)
    left["lab"] = rng.choice(["p", "q"], n_left)
    left.index = rng.permutation(n_left) + 1000  # labels, not positions
    right = pd.DataFrame({"a": rng.uniform(-1, 5, n_right), "b": rng.choice(levels["b"], n_right) + rng.normal(0, 0.05, n_right),
                          "c": rng.choice([10.0, 15.0, 20.0, 30.0], n_right),  # 15.0 is an exact tie between 10 and 20
                          "cat": rng.choice(["x", "y", "z", "unknown"], n_right), "lab": rng.choice(["p", "q"], n_right)},
                         index=np.arange(n_right) + 7)
    right.loc[right.index[0], "a"] = 0.25  # another exact tie (0.0 / 0.5)
    params = [P("a", True), P("b", True), P("c", True), P("cat", False), P("lab", False), P("cont", True, discrete=False)]
    want, want_multi = reference_match(left, right, params)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        got = fuzzy_row_match(left, right, params)
    assert got.tolist() == want.tolist()
    multi = [w for w in rec if issubclass(w.category, SearchSpaceMatchWarning)]
    assert bool(multi) == bool(want_multi)
    if multi:
        assert multi[0].message.data.index.tolist() == want_multi


def test_product_space_measurements_find_their_rows_and_the_index_is_reusable():
    vals = np.arange(10) / 9.0
    grid = np.stack(np.meshgrid(vals, vals, vals, indexing="ij"), -1).reshape(-1, 3)
    left = pd.DataFrame(grid, columns=["x0", "x1", "x2"])
    params = [P(c, True) for c in left.columns]
    matcher = FuzzyRowMatcher(left, params)
    rng = np.random.default_rng(0)
    rows = rng.choice(len(left), 25, replace=False)
    noisy = left.iloc[rows] + rng.normal(0, 0.01, (25, 3))  # measurement noise well inside half a grid step
    assert matcher.match(noisy).tolist() == rows.tolist()
    assert matcher.match(left.iloc[[5, 5, 999]]).tolist() == [5, 5, 999]
    with pytest.raises(ValueError, match="right dataframe"):
        matcher.match(noisy[["x0", "x1"]])
    with pytest.raises(ValueError, match="left dataframe"):
        fuzzy_row_match(left[["x0"]], noisy, params)
    assert fuzzy_row_match(left, noisy.iloc[:0], params).tolist() == []
    empty_nan = noisy.iloc[:2].copy()
    empty_nan.iloc[0, 1] = np.nan  # NaN never equals the minimum difference: no match, row dropped
    assert matcher.match(empty_nan).tolist() == [rows[1]]


def test_frame_content_hash_sees_every_edit_without_materialising_the_frame():
    """``recommenders._frame_content_hash`` keys the resident candidate matrix: any changed value, in a single-block frame
    (built from one 2-D array) or a block-per-column frame, changes the key; equal content gives equal keys whatever the
    block layout of the copy."""
    from baybe_amd.recommenders import _frame_content_hash

    rng = np.random.default_rng(0)
    arr = rng.integers(0, 11, size=(5000, 7)) / 10.0
    single = pd.DataFrame(arr.copy(), columns=[f"x{i}" for i in range(7)])
    per_col = pd.DataFrame({c: single[c].to_numpy().copy() for c in single.columns})
    for frame in (single, per_col):
        key = _frame_content_hash(frame)
        assert key == _frame_content_hash(frame)  # (equal keys imply equal content; a copy in another memory layout may
        #                                           key differently, which only costs one spurious re-upload)
        for r, c in ((0, 0), (4999, 6), (1234, 3)):
            edited = frame.copy()
            edited.iloc[r, c] += 1e-12
            assert _frame_content_hash(edited) != key
    big = pd.DataFrame({f"x{i}": rng.random(150_000) for i in range(16)})  # above the thread-pool threshold
    key = _frame_content_hash(big)
    big.iloc[77_777, 9] = -1.0
    assert _frame_content_hash(big) != key
    mixed = pd.DataFrame({"a": [1.0, 2.0], "b": ["u", "v"]})
    assert _frame_content_hash(mixed) == _frame_content_hash(mixed.copy()) != _frame_content_hash(mixed.assign(b=["u", "w"]))
    assert isinstance(_frame_content_hash(pd.DataFrame(index=range(3))), int)


def test_numerical_columns_are_normalised_to_float_like_the_reference():
    """``normalize_input_dtypes`` (utils/dataframe.py:418-419, 745-795): a numerical column that arrives as integers or as
    strings (read back from a CSV) still matches; non-integer dtypes raise the reference's warning."""
    import warnings

    from baybe_amd.dataframe import fuzzy_row_match

    class P:
        def __init__(self, name, num):
            self.name, self.is_numerical, self.is_discrete = name, num, True

    left = pd.DataFrame({"x": [0.0, 1.0, 2.0, 0.0, 1.0, 2.0], "c": ["a", "a", "a", "b", "b", "b"]})
    params = [P("x", True), P("c", False)]
    as_int = pd.DataFrame({"x": [2, 0], "c": ["b", "a"]})
    assert list(fuzzy_row_match(left, as_int, params)) == [5, 0]
    as_str = pd.DataFrame({"x": ["1", "2.0"], "c": ["a", "b"]})
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        assert list(fuzzy_row_match(left, as_str, params)) == [1, 5]
    assert any("unexpected data types" in str(w.message) for w in rec)


def test_measurements_key_of_the_fit_cache():
    """``surrogates._frame_hash`` decides "unchanged context -> no refit" (surrogates/base.py:418-424): equal content gives equal keys
    whatever the memory layout or the index; any edited value, an added row, a renamed column or a changed dtype gives another key."""
    from baybe_amd.surrogates import _frame_hash

    rng = np.random.default_rng(0)
    df = pd.DataFrame({"a": rng.random(50), "b": rng.integers(0, 5, 50), "lab": rng.choice(["u", "v", "w"], 50), "y": rng.random(50)})
    key = _frame_hash(df)
    assert key == _frame_hash(df.copy()) == _frame_hash(df.reset_index(drop=True).set_index(pd.Index(range(100, 150))))
    block = pd.DataFrame(np.asfortranarray(df[["a", "y"]].to_numpy()), columns=["a", "y"])  # another memory layout, same content
    assert _frame_hash(block) == _frame_hash(pd.DataFrame({"a": df["a"].to_numpy(), "y": df["y"].to_numpy()}))
    for edited in (df.assign(a=df["a"].where(df.index != 7, 0.123)), df.assign(lab=df["lab"].where(df.index != 3, "z")),
                   pd.concat([df, df.iloc[:1]]), df.rename(columns={"y": "z"}), df.astype({"b": "float64"}), df.iloc[:-1]):
        assert _frame_hash(edited) != key
    nan = df.assign(y=df["y"].where(df.index != 0, np.nan))
    assert _frame_hash(nan) == _frame_hash(nan.copy()) != key
    assert _frame_hash(pd.DataFrame()) == _frame_hash(pd.DataFrame()) != key  # (a tuple of integers: process-independent)
