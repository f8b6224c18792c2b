"""fit.HostGcPacer: the cyclic collector is off inside a step loop and back (with everything unfrozen) afterwards, also after an exception."""
import gc

import pytest


def test_pacer_pauses_and_restores_the_collector(monkeypatch):
    from ppsurf_amd.fit import HostGcPacer
    monkeypatch.delenv('PPS_HOST_GC', raising=False)
    assert gc.isenabled()
    with HostGcPacer(every=2) as p:
        assert not gc.isenabled() and gc.get_freeze_count() > 0
        for _ in range(5):
            p.tick()                                  # young-generation collections only; must not re-enable anything
        assert not gc.isenabled()
    assert gc.isenabled() and gc.get_freeze_count() == 0
    with pytest.raises(RuntimeError):
        with HostGcPacer():
            raise RuntimeError('step failed')
    assert gc.isenabled() and gc.get_freeze_count() == 0
    p = HostGcPacer()
    p.close()                                         # closing what was never entered is a no-op
    assert gc.isenabled()


def test_pacer_opt_out_and_nesting(monkeypatch):
    from ppsurf_amd.fit import HostGcPacer
    monkeypatch.setenv('PPS_HOST_GC', 'auto')
    with HostGcPacer():
        assert gc.isenabled()                         # left on its automatic schedule
    monkeypatch.delenv('PPS_HOST_GC')
    with HostGcPacer():
        with HostGcPacer() as inner:                  # an inner loop finds the collector already off and leaves it to the outer one
            assert not inner.active
        assert not gc.isenabled()
    assert gc.isenabled()
