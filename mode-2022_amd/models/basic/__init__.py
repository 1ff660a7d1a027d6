from .spherical_conv import SphereConv
