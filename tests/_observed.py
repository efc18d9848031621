"""Bounds that are "twice what was observed" (VERDICT r03, item 1b): tests/golden/bf16_observed.json holds, per test case and tensor, the error
measured on an MI355X box when the case was recorded; a later run asserts value <= max(2 x recorded, 1e-4).  The engine's kernels are bit-reproducible
and the oracle runs on the same torch-CPU build, so a recorded value is reproduced to the digit on another box of the pool; the factor of two
is the whole margin.  Recording: FACEOFF_RECORD_OBSERVED=<path> python -m pytest ... writes the values instead of asserting them (the file is
then copied to tests/golden/bf16_observed.json and committed, with the log it came from under gpurun_out/)."""
import json
import os

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bf16_observed.json")
_REC = os.environ.get("FACEOFF_RECORD_OBSERVED")
FLOOR = 1e-4


def _load(path):
    if os.path.exists(path):
        with open(path) as fh:
            return json.load(fh)
    return {}


class Observed:
    def __init__(self, case):
        self.case = case
        self.table = _load(_PATH).get(case, {})
        self.new = {}

    def check(self, key, value, cap=None):
        """value <= 2 x the recorded value of (case, key) -- and <= cap, an absolute ceiling that holds whatever was recorded."""
        value = float(value)
        self.new[key] = value
        if cap is not None:
            assert value <= cap, (self.case, key, value, cap)
        if _REC:
            return
        assert key in self.table, f"no recorded value for {self.case}/{key}: record with FACEOFF_RECORD_OBSERVED"
        # FLOOR: errors below 1e-4 (1/40 of one bf16 rounding, 2^-8) are fp32 summation-order noise -- a bias gradient at 1e-5 moves by 3x
        # when a kernel changes its accumulation order; twice-the-recorded-value only means something above that floor
        assert value <= max(2.0 * self.table[key], FLOOR), (self.case, key, value, "recorded", self.table[key])

    def flush(self):
        if _REC:
            allv = _load(_REC)
            allv.setdefault(self.case, {}).update(self.new)
            os.makedirs(os.path.dirname(os.path.abspath(_REC)), exist_ok=True)
            with open(_REC, "w") as fh:
                json.dump(allv, fh, indent=0, sort_keys=True)
