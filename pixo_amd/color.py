"""`pixo::ColorType` (reference src/color.rs:7-31)."""
import enum


class ColorType(enum.IntEnum):
    Gray = 0
    GrayAlpha = 1
    Rgb = 2
    Rgba = 3

    def bytes_per_pixel(self) -> int:
        return (1, 2, 3, 4)[int(self)]

    @classmethod
    def try_from(cls, value: int) -> "ColorType":
        """TryFrom<u8> (color.rs:85-98): raises ValueError(value) for unknown discriminants."""
        if value not in (0, 1, 2, 3):
            raise ValueError(value)
        return cls(value)
