from .factories import Conv, Pool
