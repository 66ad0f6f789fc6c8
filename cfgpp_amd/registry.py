"""Name -> class table behind ``register_solver`` / ``get_solver`` (reference: latent_diffusion.py:13-26,
latent_sdxl.py:15-28: a module-level dict, ``ValueError`` for a duplicate registration and for an unknown name).
One implementation shared by the SD1.5 and SDXL solver modules; insertion order = registration order."""
from __future__ import annotations


class Registry(dict):
    def __init__(self, what: str = "Solver"):
        super().__init__()
        self.what = what

    def register(self, name: str):
        """``@register(name)`` class decorator."""
        def add(cls):
            if name in self:
                raise ValueError(f"{self.what} {name} already registered.")
            self[name] = cls
            return cls
        return add

    def create(self, name: str, **kwargs):
        """Instantiate the class registered under ``name`` with ``kwargs``."""
        try:
            cls = self[name]
        except KeyError:
            raise ValueError(f"{self.what} {name} does not exist.") from None
        return cls(**kwargs)
