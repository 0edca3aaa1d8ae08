"""Constants and enums of the reference's public API (oinkoink/utils.py:4-34), restated."""
from enum import Enum, IntEnum


class Connect4Stats:
    height = 6
    width = 7
    area = 42


class Side(IntEnum):
    o = 0
    x = 1

    @classmethod
    def as_str(cls, side):
        return "o" if side == cls.o else "x"


class Result(Enum):
    o_win = 1.0
    x_win = 0.0
    draw = 0.5

    # A drop-in's results meet the reference's own enum in the caller's code (training.py:137-141 counts
    # `oinkoink.utils.Result.o_win` in a list of ours): members compare equal across the two classes when
    # name and value agree.  Enum's default is identity, which would count 0, 0, 0.
    def __eq__(self, other):
        if self is other:
            return True
        return (isinstance(other, Enum) and type(other).__name__ == "Result" and other.name == self.name
                and other.value == self.value)

    def __ne__(self, other):
        return not self.__eq__(other)

    def __hash__(self):
        return hash(self.name)


RESULT_FROM_CODE = {-1: None, 0: Result.x_win, 1: Result.draw, 2: Result.o_win}
CODE_FROM_RESULT = {None: -1, Result.x_win: 0, Result.draw: 1, Result.o_win: 2}


def same_side(result, side):
    return (result == Result.o_win and side == Side.o) or (result == Result.x_win and side == Side.x)


def value_to_side(value, side):
    return value if side == Side.o else 1.0 - value
