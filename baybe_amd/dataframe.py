"""Search-space row matching for large discrete spaces (last item).

``Campaign.add_measurements`` marks measured rows through ``fuzzy_row_match``
(``baybe/campaign.py:366-370`` -> ``baybe/utils/dataframe.py:361-460``), which materialises, per parameter, a
``len(right) x len(left)`` comparison / absolute-difference matrix: at 1e6 search-space rows and 20 parameters that is
seconds and gigabytes for a handful of measurements, and it is paid on every call.  The semantics, however, are
column-separable:

* a categorical column matches where the values are equal;
* a numerical column matches where ``|right - left|`` equals the minimum of that quantity over **all** left rows - i.e.
  where the left value is (one of) the value(s) of that column nearest to the right value, independently of the other
  columns;
* a left row matches if every column matches; the first matching row (left order) is returned per right row, right rows
  without a match are dropped, several matches raise a warning.

So every left row can be reduced once to a 64-bit key over its per-column value codes (cached per search space: O(N p)
once), and a right row to the key(s) of its nearest / equal codes (two per numerical column only in an exact tie);
matching is then a sorted-key lookup with an exact check of the codes, O(k p log N) per call.  Same signature and result as the reference
function; ``FuzzyRowMatcher`` keeps the index for repeated calls against the same left frame.
"""

from __future__ import annotations

import itertools
import warnings
from typing import Sequence

import numpy as np
import pandas as pd


class SearchSpaceMatchWarning(UserWarning):
    """``baybe.exceptions.SearchSpaceMatchWarning``: several search-space rows match one input row."""

    def __init__(self, message: str, data: pd.DataFrame):
        super().__init__(message)
        self.data = data


def _split_columns(parameters):
    cat = [p.name for p in parameters if (not p.is_numerical and p.is_discrete)]
    num = [p.name for p in parameters if (p.is_numerical and p.is_discrete)]
    other = {p.name for p in parameters if not p.is_discrete}
    provided = {p.name for p in parameters}
    assert set(cat) | set(num) | other == provided, (
        f"There are parameter types that would be silently ignored: {provided.difference(set(cat) | set(num) | other)}"
    )
    return cat, num


def _numeric_values(df: pd.DataFrame, col: str) -> np.ndarray:
    """Column of a numerical parameter as float64 - the ``normalize_input_dtypes`` step of the reference
    (``utils/dataframe.py:418-419, 745-795``): integer columns are converted silently, anything else that is not already float
    (strings read back from a CSV, objects) with the reference's warning.  Categorical columns are compared as they are, in
    the reference as here."""
    ser = df[col]
    if not pd.api.types.is_float_dtype(ser) and not pd.api.types.is_integer_dtype(ser):
        warnings.warn(f"The following columns have unexpected data types: {[col]}. Converting to float internally.", UserWarning)
    return np.asarray(ser, dtype=np.float64)


class FuzzyRowMatcher:
    """Index over the rows of ``left_df`` for repeated ``fuzzy_row_match`` calls with the same parameters."""

    def __init__(self, left_df: pd.DataFrame, parameters: Sequence):
        self.cat_cols, self.num_cols = _split_columns(parameters)
        if diff := (set(self.cat_cols) | set(self.num_cols)).difference(left_df.columns):
            raise ValueError(
                f"For fuzzy row matching, all discrete parameters need to have a corresponding column in the left "
                f"dataframe. Parameters not found: {diff})"
            )
        self.index = left_df.index
        self.levels: dict[str, np.ndarray] = {}
        cols = self.cat_cols + self.num_cols
        # Per-column value codes of every left row, and a 64-bit multiply-add hash of the code tuple as the row key
        # (a mixed-radix key would overflow: 11^20 rows-worth of key space for a 20-parameter grid).  Hash collisions can
        # only add candidates to a key's range - every candidate is checked against the codes - never lose a match.
        self._codes = np.zeros((len(left_df), len(cols)), dtype=np.int32)
        self._mult = (np.random.default_rng(0x5EED).integers(1, 2**63, size=len(cols), dtype=np.uint64) << np.uint64(1)) | np.uint64(1)
        keys = np.zeros(len(left_df), dtype=np.uint64)
        for ci, col in enumerate(cols):
            values = _numeric_values(left_df, col) if col in self.num_cols else np.asarray(left_df[col])
            levels, codes = np.unique(values, return_inverse=True)  # sorted levels: nearest-value search for numbers
            self.levels[col] = levels
            self._codes[:, ci] = codes
            keys += codes.astype(np.uint64) * self._mult[ci]  # wraps modulo 2^64
        self._order = np.argsort(keys, kind="stable")  # stable: the first left row of a key stays first
        self._sorted = keys[self._order]

    def _codes_for(self, col: str, values: np.ndarray) -> list[np.ndarray]:
        """Per right value the left codes that match it in this column (empty: none)."""
        levels = self.levels[col]
        out = []
        if col in self.num_cols:
            v = np.asarray(values, dtype=np.float64)
            if len(levels) == 0:
                return [np.empty(0, dtype=np.int64) for _ in v]
            pos = np.clip(np.searchsorted(levels, v), 1, max(len(levels) - 1, 1))
            for x, p in zip(v, pos):
                cand = np.unique(np.clip([p - 1, p], 0, len(levels) - 1))
                diff = np.abs(x - levels[cand])
                out.append(cand[diff == diff.min()].astype(np.int64))  # both neighbours in an exact tie; NaN: none
        else:
            lookup = {lv: i for i, lv in enumerate(levels.tolist())}
            for x in np.asarray(values).tolist():
                out.append(np.array([lookup[x]], dtype=np.int64) if x in lookup else np.empty(0, dtype=np.int64))
        return out

    def match(self, right_df: pd.DataFrame) -> pd.Index:
        if diff := (set(self.cat_cols) | set(self.num_cols)).difference(right_df.columns):
            raise ValueError(
                f"For fuzzy row matching, all discrete parameters need to have a corresponding column in the right "
                f"dataframe. Parameters not found: {diff})"
            )
        cols = self.cat_cols + self.num_cols
        per_col = [self._codes_for(c, _numeric_values(right_df, c) if c in self.num_cols else right_df[c].to_numpy()) for c in cols]
        matched, multiple = [], []
        for r in range(len(right_df)):
            options = [per_col[ci][r] for ci in range(len(cols))]
            first, count = None, 0
            if all(len(o) for o in options):
                for combo in itertools.product(*options):  # one combination unless a numerical value sits in an exact tie
                    code = np.asarray(combo, dtype=np.int32)
                    key = np.uint64(0)
                    with np.errstate(over="ignore"):
                        for cv, mv in zip(code, self._mult):
                            key = key + np.uint64(cv) * mv
                    lo, hi = np.searchsorted(self._sorted, key, "left"), np.searchsorted(self._sorted, key, "right")
                    rows = self._order[lo:hi]
                    rows = rows[(self._codes[rows] == code[None, :]).all(axis=1)]  # drop hash collisions
                    if len(rows):
                        count += len(rows)
                        pos = int(rows.min())
                        first = pos if first is None or pos < first else first
            elif not cols and len(self.index):  # no discrete parameter at all: every left row matches
                first, count = 0, len(self.index)
            if first is not None:
                matched.append(first)
                if count > 1:
                    multiple.append(right_df.index[r])
        if multiple:
            warnings.warn(SearchSpaceMatchWarning(
                f"Some input rows have multiple matches with the search space. Matching only first occurrence for these "
                f"rows. Indices with multiple matches: {multiple}", right_df.loc[multiple]))
        return pd.Index(self.index[np.asarray(matched, dtype=np.int64)]) if matched else pd.Index([], dtype=self.index.dtype)


def fuzzy_row_match(left_df: pd.DataFrame, right_df: pd.DataFrame, parameters: Sequence) -> pd.Index:
    """Drop-in for ``baybe.utils.dataframe.fuzzy_row_match`` (``utils/dataframe.py:361-460``): the index of the rows of
    ``left_df`` matched by the rows of ``right_df``.  Builds the row index on every call; keep a ``FuzzyRowMatcher`` per
    search space to pay for it once."""
    return FuzzyRowMatcher(left_df, parameters).match(right_df)


def add_parameter_noise(data: pd.DataFrame, parameters: Sequence, noise_type: str = "absolute", noise_level: float = 1.0) -> pd.DataFrame:
    """``baybe.utils.dataframe.add_parameter_noise`` (utils/dataframe.py:124-175): uniform noise on the values of the numerical
    parameters of a recommendation frame - additive in [-level, level] ("absolute") or multiplicative in [1 - level/100, 1 +
    level/100] ("relative_percent") - drawn from numpy's global generator, one draw per parameter and row; continuous
    parameters are clipped to their bounds.  Changes ``data`` in place and returns it."""
    if noise_type not in ("relative_percent", "absolute"):
        raise ValueError(f"Parameter 'noise_type' was {noise_type} but must be either 'absolute' or 'relative_percent'.")
    for prm in parameters:
        if not prm.is_numerical:
            continue
        n = len(data)
        if noise_type == "absolute":
            data[prm.name] = data[prm.name] + np.random.uniform(-noise_level, noise_level, n)
        else:
            data[prm.name] = data[prm.name] * np.random.uniform(1.0 - noise_level / 100.0, 1.0 + noise_level / 100.0, n)
        if getattr(prm, "is_continuous", False):
            data[prm.name] = data[prm.name].clip(prm.bounds.lower, prm.bounds.upper)
    return data
