"""`pixo::Error` (reference src/error.rs:10-91): one exception class per variant this
path can raise; str(e) is the reference's Display string."""


class Error(Exception):
    """Base of all pixo errors (the `pixo::Error` enum)."""


class InvalidDimensions(Error):
    pass


class InvalidDataLength(Error):
    pass


class InvalidQuality(Error):
    pass


class ImageTooLarge(Error):
    pass


class UnsupportedColorType(Error):
    pass


class CompressionError(Error):
    pass


class InvalidRestartInterval(Error):
    pass


class InvalidColorArgument(Error):
    """wasm.rs:122-131: the flat entry's own colour check (JsError in the reference)."""


class BufferTooSmall(Error):
    pass


_BY_STATUS = {
    -1: InvalidDimensions, -2: InvalidDataLength, -3: InvalidQuality, -4: ImageTooLarge,
    -5: UnsupportedColorType, -6: CompressionError, -7: InvalidRestartInterval,
    -8: InvalidColorArgument, -9: BufferTooSmall,
}


def from_status(status: int, message: str) -> Error:
    return _BY_STATUS.get(status, Error)(message)
