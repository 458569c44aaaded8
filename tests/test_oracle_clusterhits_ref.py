"""clusterhits oracle pinned to the reference's own code: oracle/sd_oracle.cpp's restatement against
oracle/_ref/libsdref_ch.so -- R/src/util/ClusterHits.cpp compiled where it lies (scores, compatibility, P-values, logGamma
are the reference's functions; the merge loop of :363-485 is re-driven over them) -- on 1 500 synthetic entries:
identical partition, printed member order and P-value bit patterns."""
import numpy as np
import pytest

from oracle.pyoracle import oracle_clusterhits, ref_ch_available, RefClusterHits
from chgen import many_entries


@pytest.mark.skipif(not ref_ch_available(), reason='oracle/_ref/libsdref_ch.so not built (needs /root/reference)')
def test_oracle_clusterhits_equals_reference_functions(oracle):
    ref = RefClusterHits()
    # the logGamma table of ClusterHits.cpp:267-271 from the reference's logGamma == the product's host table
    from spacedust_amd.api import Host
    assert (Host().lgamma_table(5000) == ref.lgamma_table(5000)).all()
    n_clusters = n_entries = 0
    for (q, t, s, p, genome) in many_entries(2024, 1500):
        for cls, d in ((2, 3), (3, 1)) if n_entries % 10 == 0 else ((2, 3),):
            cof, mo, cs, pco, pmh, _ = oracle_clusterhits(oracle, q, t, s, p, genome, d=d, cls=cls)
            rcof, rrank, rcs, rpco, rpmh = ref.entry(q, t, s, p, genome, d=d, cls=cls)
            assert (cof == rcof).all() and (cs == rcs).all()
            assert pco.tobytes() == rpco.tobytes() and pmh.tobytes() == rpmh.tobytes()
            w = 0
            for c in range(len(cs)):
                assert (rrank[mo[w:w + cs[c]]] == np.arange(cs[c])).all()
                w += cs[c]
            n_clusters += len(cs)
        n_entries += 1
    assert n_entries == 1500 and n_clusters > 1000


@pytest.mark.skipif(not ref_ch_available(), reason='oracle/_ref/libsdref_ch.so not built (needs /root/reference)')
def test_host_cluster_pvalues_equal_reference_functions():
    """sd_host_cluster_pvalues (csrc/host/sd_chpval.cpp: the P-values the device path attaches to every emitted cluster) against
    the reference's clusterMatchScore / multihitPval on the clusters of 600 synthetic entries: bit patterns and member order"""
    import ctypes as C
    from spacedust_amd import _lib
    from spacedust_amd.api import Host, ptr
    L = _lib.load()
    ref = RefClusterHits()
    lg = Host().lgamma_table(5000)
    n_clusters = 0
    for (q, t, s, p, genome) in many_entries(77, 600):
        for alpha in (1.0, 0.01):
            rcof, rrank, rcs, rpco, rpmh = ref.entry(q, t, s, p, genome, alpha=alpha, p_clu=1.0, p_mh=1.0)
            for c in range(len(rcs)):
                m = np.nonzero(rcof == c)[0]
                m = m[np.argsort(rrank[m])][::-1].copy()   # any input order
                co, mh = C.c_double(), C.c_double()
                order = np.zeros(len(m), np.uint32)
                qq, tt, ss, pp = (np.ascontiguousarray(a[m]) for a in (q, t, s, p))
                assert L.sd_host_cluster_pvalues(len(m), ptr(qq), ptr(tt), ptr(ss), ptr(pp), genome, alpha, ptr(lg), len(lg), C.byref(co),
                                                 C.byref(mh), ptr(order)) == 0
                assert np.float64(co.value).tobytes() == rpco[c:c + 1].tobytes(), (co.value, rpco[c])
                assert np.float64(mh.value).tobytes() == rpmh[c:c + 1].tobytes(), (mh.value, rpmh[c])
                assert (rrank[m[order]] == np.arange(len(m))).all()
                n_clusters += 1
    assert n_clusters > 800
