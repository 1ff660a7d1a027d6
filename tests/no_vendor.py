"""The guard against silent vendor-library exits lives in the package (mode_hip/no_vendor.py: bench.py uses it too); the tests import it
from here as before."""
from mode_hip.no_vendor import FORBIDDEN, no_vendor_arithmetic  # noqa: F401
