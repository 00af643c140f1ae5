"""Drawing process of vapor_amd.figures: reads pickled figure specifications from stdin (8-byte length, payload; length 0
ends it), draws each with matplotlib (figures.render) and answers one byte per figure - 0, or 1 followed by the length
and text of what the drawing raised.  It never loads the HIP library: a fresh interpreter with numpy and matplotlib."""
import pickle
import struct
import sys
import traceback


def main() -> int:
    from vapor_amd import figures
    inp, out = sys.stdin.buffer, sys.stdout.buffer
    while True:
        head = inp.read(8)
        if len(head) < 8:
            return 0
        n = struct.unpack("<q", head)[0]
        if n <= 0:
            return 0
        payload = inp.read(n)
        try:
            figures.render(pickle.loads(payload))
            out.write(b"\x00")
        except Exception:       # noqa: BLE001 - reported to the parent, which raises it
            msg = traceback.format_exc().encode()
            out.write(b"\x01" + struct.pack("<q", len(msg)) + msg)
        out.flush()


if __name__ == "__main__":
    raise SystemExit(main())
