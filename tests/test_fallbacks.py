"""Loud fallbacks (round-4 verdict, weak item 1): a CUDA forward that leaves the hand-written kernels for the torch modules (vendor
libraries) warns once per (place, reason), raises under OCTA_STRICT=1, and is silent only where the torch modules are asked for."""
import warnings

import pytest

from octa_autosegmentation_amd.models import networks


def test_vendor_fallback_warns_once_raises_when_strict_and_is_silent_for_references(monkeypatch):
    monkeypatch.setenv("OCTA_STRICT", "0")
    networks._WARNED.clear()
    n0 = networks.PATH_COUNTS["vendor"]
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        networks._vendor_fallback("somewhere", "some reason")
        networks._vendor_fallback("somewhere", "some reason")
        networks._vendor_fallback("somewhere", "another reason")
    assert [str(w.message).split(":")[0] for w in rec] == ["somewhere", "somewhere"]      # once per (place, reason)
    assert all(issubclass(w.category, RuntimeWarning) for w in rec)
    assert networks.PATH_COUNTS["vendor"] == n0 + 3                                       # every occurrence is counted
    monkeypatch.setenv("OCTA_STRICT", "1")
    with pytest.raises(networks.VendorFallbackError, match="some reason"):
        networks._vendor_fallback("somewhere", "some reason")
    with networks.vendor_reference():                                                     # the reference side of a parity test
        networks._vendor_fallback("somewhere", "some reason")
    monkeypatch.setattr(networks, "USE_MFMA_CONV", False)                                 # ... or the module-wide switch
    networks._vendor_fallback("somewhere", "some reason")


def test_mfma_refusal_names_the_reason():
    import torch
    net = networks.DynUNet(2, 1, 1, [3, 3, 3, 3, 3], [1, 2, 2, 2, 1], [1, 2, 2, 2, 1])
    assert "CPU tensor" in net._mfma_refusal(torch.zeros(1, 1, 32, 32))
    odd = networks.DynUNet(2, 1, 1, [3, 3, 3], [1, 2, 1], [1, 2], filters=[24, 48, 96])
    # (reasons that need a CUDA tensor are exercised in tests/test_training_cli_gpu.py)
    assert odd._mfma_refusal(torch.zeros(1, 1, 32, 32)) is not None
