"""Worker process of vapor_amd.hostpool: reads pickled (module, function, arguments) from stdin (8-byte length, payload; a
length of 0 or the end of the stream ends it), calls the function and answers with a tag byte - 0 result, 1 what the
call raised (the exception itself when it can be pickled, else its traceback text) -, an 8-byte length and the pickle.
It never loads the HIP library: a fresh interpreter with numpy and, on demand, matplotlib / sklearn / scipy."""
import importlib
import pickle
import struct
import sys
import traceback


def main() -> int:
    inp, out = sys.stdin.buffer, sys.stdout.buffer
    sys.stdout = sys.stderr                       # (nothing a library prints may land in the reply stream)
    while True:
        head = inp.read(8)
        if len(head) < 8:
            return 0
        n = struct.unpack("<q", head)[0]
        if n <= 0:
            return 0
        payload = inp.read(n)
        try:
            module, function, args = pickle.loads(payload)
            tag, body = b"\x00", pickle.dumps(getattr(importlib.import_module(module), function)(*args), protocol=4)
        except Exception as e:       # noqa: BLE001 - handed to the parent, which raises it
            tag = b"\x01"
            try:
                body = pickle.dumps(e, protocol=4)
                pickle.loads(body)
            except Exception:        # noqa: BLE001
                body = pickle.dumps(traceback.format_exc(), protocol=4)
        out.write(tag + struct.pack("<q", len(body)) + body)
        out.flush()


if __name__ == "__main__":
    raise SystemExit(main())
