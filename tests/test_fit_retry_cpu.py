"""fit()'s answer to an ENGINE timeout (VERDICT r02 Weak #8): a chain's workgroups must all be resident; when the engine's exchange
gives up (BL_ERR_TIMEOUT -> TimeoutError "biolith_hip: ..."), fit() launches once more on half the workgroups per chain, and the
caller's own time limit ("Timed out") is never retried.  Host logic only: the device handle is a stand-in."""
import contextlib
import io

import numpy as np
import pytest

import biolith_amd.engine as engine
from biolith_amd.models import occu, simulate
from biolith_amd.utils import fit


class _FakeDataset:
    log = []

    def __init__(self, site_covs, obs_covs, obs, *a, **kw):
        self.N, self.Ks = np.shape(site_covs)
        _, self.T, self.J, self.Ko = np.shape(obs_covs)
        self.D = self.Ks + self.Ko + 2
        self.S = 1
        self._k = 24

    def launch(self, num_warmup, num_samples, num_chains, seed, chain_offset, wgs_per_chain=0, **kw):
        self._shape = (num_chains, num_samples)
        self._k = wgs_per_chain or 24
        _FakeDataset.log.append(("launch", self._k))

    def done(self):
        return True

    def wgs_per_chain(self):
        return self._k

    def abort(self):
        _FakeDataset.log.append(("abort",))

    def wait(self):
        if self._k == 24 and _FakeDataset.mode == "engine":
            raise TimeoutError("biolith_hip: exchange timed out (a cooperating workgroup is not resident)")
        if _FakeDataset.mode == "caller":
            raise TimeoutError("Timed out")

    def fetch(self):
        C, S = self._shape
        z = np.zeros
        return engine.NutsResult(z((C, S, self.D), np.float32), z((C, S), bool), np.ones((C, S), np.int32), z((C, S), np.float32), z((C, S), np.float32),
                                 np.ones(C, np.float32), np.ones((C, self.D), np.float32), z((C, 2), np.int64), 1.0, self._k, 0, True)


@pytest.fixture
def fake(monkeypatch):
    monkeypatch.setattr(engine, "OccuDataset", _FakeDataset)
    _FakeDataset.log = []
    with contextlib.redirect_stdout(io.StringIO()):
        data, _ = simulate(n_sites=20, random_seed=0)
    return data


def test_engine_timeout_is_retried_once_on_half_the_workgroups(fake):
    _FakeDataset.mode = "engine"
    with pytest.warns(RuntimeWarning, match="retrying once with 12"):
        res = fit(occu, **fake, num_chains=2, num_samples=5, num_warmup=5)
    assert [e for e in _FakeDataset.log if e[0] == "launch"] == [("launch", 24), ("launch", 12)]
    assert ("abort",) in _FakeDataset.log                     # the failed launch was aborted and waited for before the retry
    assert res.mcmc.result.wgs_per_chain == 12


def test_the_callers_time_limit_is_not_retried(fake):
    _FakeDataset.mode = "caller"
    with pytest.raises(TimeoutError, match="Timed out"):
        fit(occu, **fake, num_chains=2, num_samples=5, num_warmup=5)
    assert [e for e in _FakeDataset.log if e[0] == "launch"] == [("launch", 24)]
