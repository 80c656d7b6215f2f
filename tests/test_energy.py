"""The energy model of train.py:470-515 evaluated on synthetic rate lists (CPU) and on the detector's
spike-rate mode (GPU)."""
import pytest
import torch

from snn_automotive_object_detection_amd import energy


def _fake_rates():
    rates = {}
    for i in range(15):                    # 5 levels x (shared, obj, bbox), 2 images
        rates[i] = torch.tensor([[0.05 + 0.01 * i, 1000.0 * (i + 1)], [0.07 + 0.01 * i, 1000.0 * (i + 1)]])
    for j, f in zip(range(15, 19), (12544 * 1024, 1024 * 1024, 1024 * 9, 1024 * 36)):
        rates[j] = torch.tensor([[0.08, float(f)], [0.10, float(f)], [0.12, float(f)]])
    return rates


def test_energy_report_arithmetic():
    rep = energy.energy_report(_fake_rates(), 8, 12)
    assert [l["layer"] for l in rep["layers"]] == ["LVL_0", "LVL_1", "LVL_2", "LVL_3", "pool", "FC6", "FC7"]
    l0 = rep["layers"][0]                                  # position 0: mean rate 0.06 over 2 images, T=8
    assert l0["mean_spikes"] == pytest.approx(0.06 * 8) and l0["flops"] == 1000.0
    assert l0["ann_energy_j"] == pytest.approx(1000.0 * 4.6e-12)
    assert l0["snn_energy_j"] == pytest.approx(0.48 * 1000.0 * 0.9e-12)
    fc6 = rep["layers"][5]                                 # detector: x 1000 RoIs (train.py:494), T=12
    assert fc6["mean_spikes"] == pytest.approx(0.10 * 12, rel=1e-6)
    assert fc6["flops"] == pytest.approx(12544 * 1024 * 1000.0)
    assert rep["snn_over_ann"] == pytest.approx(rep["snn_energy_j"] / rep["ann_energy_j"])
    assert rep["ann_energy_j"] == pytest.approx(sum(l["ann_energy_j"] for l in rep["layers"]))


@pytest.mark.parametrize("name", ["energy_city_T8_T12", "energy_bdd_T16_T24", "energy_small_T4_T6"])
def test_energy_report_equals_the_references_own_block(name):
    """tests/golden/energy_*.npz = what train.py:472-515 itself computes (exec-ed by oracle/make_golden.py) from the fixture's rate
    lists; the reference works in fp32 tensors, energy_report in Python floats: 1e-6 relative"""
    from oracle import fixtures as FX
    spec = FX.ENERGY_SPECS[name]
    exp = FX.load_expected(name)
    rep = energy.energy_report(FX.energy_rates(spec), spec["T_rpn"], spec["T_det"])
    assert [l["layer"] for l in rep["layers"]] == [str(x) for x in exp["layer_names"]]
    got = [[l["mean_spikes"], l["flops"]] for l in rep["layers"]]
    assert len(got) == len(exp["per_layer"])
    for g, e in zip(got, exp["per_layer"]):
        assert g[0] == pytest.approx(e[0], rel=1e-6) and g[1] == pytest.approx(e[1], rel=1e-6)
    assert rep["ann_energy_j"] == pytest.approx(float(exp["ann_total"]), rel=1e-6)
    assert rep["snn_energy_j"] == pytest.approx(float(exp["snn_total"]), rel=1e-6)
    assert rep["snn_over_ann"] == pytest.approx(float(exp["snn_total"]) / float(exp["ann_total"]), rel=1e-6)


@pytest.mark.gpu
def test_extract_spike_rates_on_the_detector(gpu_device):
    import snn_automotive_object_detection_amd as S
    torch.manual_seed(0)
    m = S.create_model("cityscapes", 9, True, True, 0, False, False, num_steps_rpn=4, num_steps_detector=6)
    m.transform.min_size, m.transform.max_size = 256, 512
    m = m.to(gpu_device).eval()
    batches = [[torch.rand((3, 256, 512), device=gpu_device)], [torch.rand((3, 256, 512), device=gpu_device)]]
    rates = energy.extract_spike_rates(m, batches)
    assert sorted(rates) == list(range(19))
    assert rates[0].shape == (2, 2) and rates[15].shape[1] == 2 and rates[15].shape[0] == rates[16].shape[0]
    assert not m.rpn.head.spike_rates and not m.roi_heads.box_head_and_predictor.spike_rates     # flags restored
    rep = energy.energy_report(rates, 4, 6)
    assert len(rep["layers"]) == 7 and 0.0 <= rep["snn_over_ann"] < 10.0
    # the shared-LIF rate is spikes / (T*C*H*W): between 0 and 1
    assert all(0.0 <= float(rates[k][:, 0].max()) <= 1.0 for k in (0, 3, 6, 9, 12, 15, 16))
