# import spherical convolution
from .sphere_conv import SphereConv
