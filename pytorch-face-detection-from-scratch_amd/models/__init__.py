from .BaseModel import BaseModel  # noqa: F401
from .ModelMeta import ModelMeta  # noqa: F401
