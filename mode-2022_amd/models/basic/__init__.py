"""Building blocks shared by the networks; re-exports the spherical convolution module class under the reference's import path
(``from models.basic import SphereConv``)."""
from .spherical_conv.sphere_conv import SphereConv

__all__ = ['SphereConv']
