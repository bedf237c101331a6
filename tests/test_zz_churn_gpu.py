"""configs[4] of BASELINE.json at full size on one GPU: 10M route keys in 1000 tenants, one batch of 100k subscribe / unsubscribe
mutations through bmq_routes_apply, then a 1M-publish batch.  The body is tests/util.py::churn_case (verified on a host-only
engine by tests/test_host.py); this file sorts last on purpose: it is the longest GPU test."""
import pytest

import bifromq_amd as B
from tests import util as U

pytestmark = pytest.mark.gpu


def test_full_size_config5_churn_then_match():
    eng = B.Engine(device=0)
    try:
        n = U.churn_case(eng, lambda tn, tt, packed: eng.match_batch(tn, tt, packed_topics=packed), n_tenants=1000, per_tenant=10_000,
                         n_ops=100_000, n_topics=1_000_000, sample_tenants=1000, n_sample=None)  # every row of the batch (rounds 3-4: of 128 tenants)
        assert 9_900_000 < n < 10_100_000
    finally:
        eng.close()
