"""
Simulation: the reference's driver object (src/var_bayes/simulation.py:17-347) on top of vgpa_amd's classes --
`setup(params, data)` builds the model, trajectory, observations and priors from the JSON parameter dictionary,
`run()` optimises the free energy with SCG, `save()` writes `<name>.h5` with one dataset per output key, and the
module-level `load(filename)` reads such a file back.  Host-side glue only; the arithmetic is the GPU sweep.

Two additions over the reference, both opt-in: `run(device_resident=True)` uses the SCG whose vectors stay in HBM
(scg.DeviceSCG), and `run(options=...)` overrides the optimiser options (the reference hard-codes them).
"""
import time
from pathlib import Path

import numpy as np

from .dynamics import dynamical_systems
from .h5io import save_h5, load_h5
from .likelihood import GaussianLikelihood, PriorKL0
from .numerics import FwdOde, BwdOde
from .scg import SCG
from .variational import VarGP

SCG_OPTIONS = {"max_it": 500, "x_tol": 1.0e-6, "f_tol": 1.0e-8, "display": True}    # simulation.py:224-225


class Simulation(object):

    __slots__ = ("name", "m_data", "output")

    def __init__(self, name: str = None) -> None:
        self.name = str(name) if name else "ID_None"
        self.m_data = {}
        self.output = {}

    @classmethod
    def _stochastic_model(cls, model: str, *args):
        key = str(model).upper()
        if key not in dynamical_systems:
            raise ValueError(f" {cls.__name__}: Unknown stochastic model -> {key}")
        return dynamical_systems[key](*args)

    def setup(self, params, data):
        """`params`: the dictionary of a sim_params_*.json file; `data`: None or (obs_t, obs_y)."""
        md = self.m_data
        md["drift"], md["noise"] = params["Drift"], params["Noise"]
        md["time_window"], md["ode_solver"] = params["Time-window"], params["Ode-method"]
        md["random_seed"], md["obs_setup"] = params["Random-Seed"], params["Observations"]
        md["mu0"], md["tau0"] = params["Prior"]["mu0"], params["Prior"]["tau0"]
        model = self._stochastic_model(params["Model"], md["noise"]["sys"], md["drift"]["theta"], md["random_seed"])
        md["model"], md["single_dim"] = model, model.single_dim
        tw = md["time_window"]
        model.make_trajectory(tw["t0"], tw["tf"], tw["dt"])
        if data is not None:
            md["obs_t"], md["obs_y"] = data[0], data[1]
        else:
            md["obs_t"], md["obs_y"], md["obs_noise"] = model.collect_obs(md["obs_setup"]["density"], md["noise"]["obs"],
                                                                          md["obs_setup"]["operator"])
        # the draw of m0 follows the trajectory and observation noise in the generator's stream (simulation.py:168-186)
        if md["single_dim"]:
            md["m0"] = model.sample_path[0] + 0.1 * model.rng.standard_normal()
            md["s0"] = 0.2
        else:
            dim_d = model.sample_path.shape[-1]
            md["m0"] = model.sample_path[0] + 0.1 * model.rng.standard_normal(dim_d)
            md["s0"] = 0.2 * np.eye(dim_d)
            md["mu0"] = md["mu0"] * np.ones(dim_d)
            md["tau0"] = md["tau0"] * np.eye(dim_d)

    def _vgpa(self, batch=1):
        md = self.m_data
        dt, method, single = md["time_window"]["dt"], md["ode_solver"], md["single_dim"]
        lik = GaussianLikelihood(md["obs_y"], md["obs_t"], md["obs_noise"], md["obs_setup"]["operator"], single)
        return VarGP(md["model"], md["m0"], md["s0"], FwdOde(dt, method, single), BwdOde(dt, method, single), lik,
                     PriorKL0(md["mu0"], md["tau0"], single), md["obs_y"], md["obs_t"], batch=batch)

    def run(self, options=None, device_resident=False):
        vgpa = self._vgpa()
        opts = dict(SCG_OPTIONS if options is None else options)
        x0 = vgpa.initialization()
        optimize = vgpa.device_scg(opts) if device_resident else SCG(vgpa.free_energy, vgpa.gradient, opts)
        t0 = time.perf_counter()
        x, fx = optimize(x0.copy())
        print(f" Elapsed time: {(time.perf_counter() - t0):.2f} seconds.", end='\n')
        if device_resident:
            vgpa.free_energy(x)                  # leave the state of the returned x behind for arg_out
        model = self.m_data["model"]
        if model.single_dim:
            dim_n = model.sample_path.size
            self.output["at"], self.output["bt"] = x[:dim_n], x[dim_n:]
        else:
            dim_n, dim_d = model.sample_path.shape
            dim_tot = dim_n * dim_d * dim_d
            self.output["at"] = x[:dim_tot].reshape(dim_n, dim_d, dim_d)
            self.output["bt"] = x[dim_tot:].reshape(dim_n, dim_d)
        self.output["fx"] = fx
        self.output.update(vgpa.arg_out)

    def save(self):
        """One dataset per key of `output` in `<name>.h5` (spaces -> underscores), scalars as shape-(1,) arrays."""
        if not self.output:
            print(f" {self.__class__.__name__}: Simulation data structure 'output' is empty.")
            return
        print(f" Saving the results to: {self.name}.h5")
        save_h5(Path(self.name.strip().replace(" ", "_") + ".h5"), self.output)


def load(filename=None):
    """Dictionary of every dataset of a result file written by `Simulation.save` (or by the reference's)."""
    if filename is None:
        raise RuntimeError(" load_data: No input file is given.")
    return load_h5(Path(filename))
