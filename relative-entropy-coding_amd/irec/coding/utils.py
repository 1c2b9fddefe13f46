"""rec/coding/utils.py: CodingError (:6) and stateless_gumbel_sample (:9-12, as the host entry point behind
ImportanceSampler's alpha < inf branch)."""
from ..errors import CodingError  # noqa: F401
