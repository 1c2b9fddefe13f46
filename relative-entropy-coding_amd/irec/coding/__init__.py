"""Mirror of rec/coding/__init__.py:1-2 of the reference."""
from .utils import CodingError  # noqa: F401
from .coder import Coder, GaussianCoder  # noqa: F401
from .beam_search_coder import BeamSearchCoder  # noqa: F401
from .samplers import Sampler, ImportanceSampler  # noqa: F401
