"""simplyp_amd -- MI355X-native time-stepping engine for the SimplyP catchment model.

Keeps the reference package's public names (``Current_Release/v0-2A/simplyP/__init__.py:1-31``):
``import simplyp_amd as sp`` then ``sp.inputs.read_input_data(...)`` / ``sp.run_simply_p(...)``
as in the reference notebooks.  ``goodness_of_fit_stats`` is mirrored (and reduced on the device for ensembles);
the plotting functions of the reference's visualise_results.py are outside the scope of this engine (SURVEY.md
section 2, row 16): the names exist and raise NotImplementedError.
"""

from . import helper_functions, inputs, marshal, abi, visualise_results          # noqa: F401
from .model import (f_x, discretized_soilP, ode_f, run_simply_p, run_simply_p_ensemble,   # noqa: F401
                    derived_P_species, sum_to_waterbody)
from .inputs import read_input_data, snow_hydrol_inputs, daily_PET      # noqa: F401
from .helper_functions import UC_Q, UC_Qinv, UC_C, UC_Cinv, UC_V, lin_interp   # noqa: F401
from .visualise_results import (plot_snow, plot_terrestrial, plot_in_stream, plot_instream_summed,   # noqa: F401
                                goodness_of_fit_stats)

__all__ = [
    'f_x', 'discretized_soilP', 'ode_f', 'run_simply_p', 'run_simply_p_ensemble',
    'derived_P_species', 'sum_to_waterbody',
    'read_input_data', 'snow_hydrol_inputs', 'daily_PET',
    'UC_Q', 'UC_Qinv', 'UC_C', 'UC_Cinv', 'UC_V', 'lin_interp',
    'plot_snow', 'plot_terrestrial', 'plot_in_stream', 'plot_instream_summed', 'goodness_of_fit_stats',
]
