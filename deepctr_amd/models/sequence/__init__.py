from .din import DIN
