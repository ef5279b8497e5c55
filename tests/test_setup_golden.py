"""The set-up functions of physicl/light.py (wavelength <-> energy, the Planck density and its binned sampler, the per-object
photon generators) against what the reference itself returned (tests/golden/make_golden.py g7_setup), seed for seed.  Host
code only: no GPU."""
import numpy as np
import pytest

import physicl as phys
import physicl.light as light
import physicl_amd.light as amd_light


def num(x):
    return np.asarray(x, dtype=np.float64)


@pytest.fixture()
def z(golden):
    return golden("g7_setup")


def test_wavelength_and_energy(z):
    assert np.array_equal([num(light.E_from_wavelength(x)) for x in z["lam"]], z["E_of_lam"])
    assert np.array_equal([num(light.wavelength_from_E(x)) for x in z["E_of_lam"]], z["lam_of_E"])


def test_planck_density_bit_exact(z):
    got = np.array([[num(light.planck_distribution(np.double(e), np.double(t))) for t in z["pd_T"]] for e in z["pd_E"]])
    assert np.array_equal(got, z["pd"])
    m = light.planck_distribution(phys.Measurement(np.double(3e-19), "J**1"), phys.Measurement(np.double(5778.0), "K**1"))
    assert np.array_equal(num(m), z["pd_measurement_args"])


def test_planck_bin_masses(z):
    lo, hi, T, bins = float(z["pp_lo"]), float(z["pp_hi"]), float(z["pp_T"]), int(z["pp_bins"])
    grid = np.linspace(lo, hi, bins)
    got = np.array([light.planck_probability(grid[k], grid[k + 1], T)[0] for k in range(bins - 1)])
    assert np.array_equal(got, z["pp_mass"])                 # the same scipy.quad of the same density


def test_planck_sampler_draw_for_draw(z):
    """400 draws after np.random.seed: the same bin for every draw, None where the reference returns None (a draw under the
    first bin's mass falls through its loop, light.py:100-103), and the np.random stream left where the reference leaves it."""
    lo, hi, T, bins = float(z["pp_lo"]), float(z["pp_hi"]), float(z["pp_T"]), int(z["pp_bins"])
    np.random.seed(int(z["ppd_seed"]))
    draws = [light.planck_phot_distribution(lo, hi, T, bins) for _ in range(len(z["ppd_E"]))]
    assert np.array_equal(np.array([d is None for d in draws]), z["ppd_none"])
    assert np.array_equal(np.array([np.nan if d is None else num(d) for d in draws]), z["ppd_E"], equal_nan=True)
    assert np.random.random() == float(z["ppd_next_random"])
    assert np.array_equal(np.asarray(amd_light._planck_cache["cdf"]), z["ppd_cdf"])      # the table itself, bit for bit
    assert z["ppd_none"].sum() < len(z["ppd_E"]) // 4


def test_generate_photons_seed_for_seed(z):
    np.random.seed(int(z["gp_seed"]))
    ph = light.generate_photons(64, min=light.E_from_wavelength(700e-9), max=light.E_from_wavelength(200e-9))
    assert np.array_equal([float(num(p.E)) for p in ph], z["gp_E"])
    assert np.array_equal(np.array([num(p.v) for p in ph]), z["gp_v"]) and np.array_equal(np.array([num(p.r) for p in ph]), z["gp_r"])
    assert np.random.random() == float(z["gp_next_random"])
    np.random.seed(int(z["gp_user_seed"]))
    ph = light.generate_photons(32, fn=lambda: np.random.random() ** 2, min=1.0, max=3.0)
    assert np.array_equal([float(num(p.E)) for p in ph], z["gp_user_E"])
    ph = light.generate_photons_from_E([np.double(1.5), np.double(2.5e-19)])
    assert np.array_equal([float(num(p.E)) for p in ph], z["gpe_E"]) and np.array_equal(np.array([num(p.v) for p in ph]), z["gpe_v"])


def test_bulk_sampler_gives_generate_photons_energies(z):
    """generate_photons_bulk(fn_vec=np.random.power) after the same seed: photon i's energy is the reference's photon i's
    (host side of PhotonBatch: no device needed to draw them)."""
    from physicl_amd import core
    np.random.seed(int(z["gp_seed"]))
    b = light.generate_photons_bulk(64, min=light.E_from_wavelength(700e-9), max=light.E_from_wavelength(200e-9),
                                    fn_vec=lambda size: np.random.power(3, size))
    assert isinstance(b, core.PhotonBatch)
    (off, E), = list(b.host_energies(0, 64))
    assert off == 0 and np.array_equal(E, z["gp_E"])
