"""Input-image contract of the pose models (reference `src/dsnt/data.py:19-36`).

Only `ImageSpecs` is part of the hot path's surface (`model.image_specs.size`,
`model.py:134-136, 225-227`); the MPII dataset loader is out of scope.
"""


class ImageSpecs:
    def __init__(self, size, subtract_mean, divide_stddev):
        self._size = size
        self._subtract_mean = subtract_mean
        self._divide_stddev = divide_stddev

    @property
    def size(self):
        return self._size

    @property
    def subtract_mean(self):
        return self._subtract_mean

    @property
    def divide_stddev(self):
        return self._divide_stddev
