"""GPU parity of the INTERMEDIATES of the MPC iteration, read back through qmhip_debug_read: the unprojected LQ model of every shooting
interval (A_d, B_d, b, Q, R, q, r, C, D, e), the projected stage record K1b hands to K3 (Ap, Bp, bp, Qp, Pp, Rp, qp, rp, Px, Pe, range of Pu) and the gains K, k
K3 leaves in it — entry by entry against the oracle on BASELINE.json config 2 (trot, N = 100), on a C5 instance and on C1 (stance: m = 18, two tile rows).  End-to-end agreement of x*, u* makes a
compensating error unlikely; this makes it impossible at the level of the blocks."""
import numpy as np
import pytest
import lq_record_check as LC

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["C2", "C5", "C1"])
def test_lq_and_stage_records_entrywise(blobs, oracle, name):
    from qm_control_amd import api, scenarios
    cfg = scenarios.make_config(name, batch=1)
    oracle.set_schedule(cfg["ev"][0], cfg["modes"][0]); oracle.set_target(cfg["ref_t"][0], cfg["ref_x"][0])
    r = oracle.mpc_step(cfg["t0"][0], cfg["t0"][0] + cfg["horizon"], cfg["x0"][0]); n = len(r["t"])
    nm = n + 3
    itf = api.QMInterface(blobs=blobs, max_batch=1, max_nodes=nm, max_ref_knots=2, max_events=cfg["ev"].shape[1])
    mpc = api.SqpMpc(itf)
    itf.debug_set("lq_debug", 1); assert itf.debug_get("lq_debug") == 1
    SRN, DBN = LC.SR["SR_SIZE"], LC.DBG["LQ_DBG_SIZE"]
    for skip in (20, 0):      # 20: no backward stage / rollout — the records as K1b wrote them; 0: the normal solve — K3's gains in the records
        itf.debug_set("riccati_skip", skip)
        mpc.set_problem(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"]); mpc.solve_resident(cfg["horizon"])
        stage = itf.debug_read("stage", (nm, SRN)); dbg = itf.debug_read("lqdbg", (nm, DBN))
        assert int(itf.debug_read("n_nodes", (1,), np.int32)[0]) == n
        worst = {}
        for i in range(n - 1):
            if r["ev"][i] == 1:
                continue
            for k, v in LC.check_interval(stage[i], dbg[i], oracle.node_lq(i), oracle.node_proj(i), after_riccati=(skip == 0)).items():
                worst[k] = max(worst.get(k, 0.0), v)
        bad = {k: v for k, v in worst.items() if not v <= 1e-10}
        assert not bad, (name, skip, bad, worst)
        print(name, "riccati_skip", skip, {k: "%.1e" % v for k, v in worst.items()})
    itf.debug_set("riccati_skip", 0); itf.debug_set("lq_debug", 0); assert itf.debug_get("lq_debug") == 0
    # the debug records change nothing: the solve without them returns the same solution bit for bit
    res_dbg = mpc.download()
    mpc.set_problem(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"]); mpc.solve_resident(cfg["horizon"]); res = mpc.download()
    assert np.array_equal(res["x"], res_dbg["x"]) and np.array_equal(res["u"], res_dbg["u"])
    # ... and the instrumented Riccati instance (skip = 32: cycle counters on, results intact) returns what the product instance returns
    itf.debug_set("riccati_skip", 32)
    mpc.set_problem(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"]); mpc.solve_resident(cfg["horizon"]); res_prof = mpc.download()
    itf.debug_set("riccati_skip", 0)
    assert np.array_equal(res["x"], res_prof["x"]) and np.array_equal(res["u"], res_prof["u"])
    itf.close()
