"""CPU restatement of the reference's text path -- TEST INFRASTRUCTURE ONLY.

Only tests/ may import this module; the product (metalchat_amd/csrc/text.cc) never does.

What is restated, and from where:
  gpt2_codec                     src/gpt.cc:20-100
  regexp / regexp_iterator       src/regexp.cc:34-178  (PCRE2 compiled with options = 0)
  byte_pair_encoder              include/metalchat/text/bpe.h:120-168 (merge), 249-270 (insert),
                                 284-342 (encode / decode)
  llama3 control tokens          src/reference.cc:113-127, src/bpe.cc:13-17
  token scanners                 include/metalchat/interpreter.h:60-175
  interpreter framing / loop     src/interpreter.cc:116-136, include/metalchat/interpreter.h:318-374

Pinning: the GPT-2 codec is pinned by the reference's own known answers (test/test_bpe.cc:29-55,
tests/test_text_cpu.py).  The control-token ids are pinned by test/test_bpe.cc:122-132 (128001 =
"<|end_of_text|>" behind a 128000-entry map).  The encoder's known answers in test/test_bpe.cc:58-119
need Meta's tokenizer.model, which is not in the reference tree and cannot be fetched here: for the
merge loop and the split PARITY IS UNPINNED -- the restatement below follows the source line by
line in its own words, the split is cross-checked against an independent engine (the `regex`
module instead of PCRE2), and tests/test_text_cpu.py runs the reference's known answers when
MC_LLAMA3_TOKENIZER_MODEL points at the file.
"""
from __future__ import annotations

import base64
import heapq

import regex

LLAMA3_PATTERN = (r"(?i:'s|'t|'re|'ve|'m|'ll|'d)|"
                  r"[^\r\n\p{L}\p{N}]?\p{L}+|"
                  r"\p{N}{1,3}|"
                  r" ?[^\s\p{L}\p{N}]+[\r\n]*|"
                  r"\s*[\r\n]+|"
                  r"\s+(?!\S)|"
                  r"\s+")

REGULAR, BEGIN_TEXT, END_TEXT, RESERVED, FINETUNE_RIGHT_PAD = 1, 2, 4, 8, 16
BEGIN_HEADER, END_HEADER, END_MESSAGE, END_TURN, IPYTHON = 32, 64, 128, 256, 512

LIMIT = 2 ** 31 - 1


# ------------------------------------------------------------------------------------------ gpt2_codec
def _gpt2_tables():
    keep = lambda b: 0x21 <= b <= 0x7E or 0xA1 <= b <= 0xAC or 0xAE <= b <= 0xFF
    enc, nxt = {}, 256
    for b in range(256):
        if keep(b):
            enc[b] = b
        else:
            enc[b] = nxt
            nxt += 1
    return enc, {v: k for k, v in enc.items()}


_ENC, _DEC = _gpt2_tables()


def gpt2_encode(data: bytes) -> str:
    return "".join(chr(_ENC[b]) for b in data)


def gpt2_decode(text: str) -> bytes:
    out = bytearray()
    for ch in text:
        cp = ord(ch)
        if cp > 0xFFFF:
            raise ValueError("code point above U+FFFF")  # std::wstring_convert<codecvt_utf8<char16_t>> throws
        out.append(_DEC[cp] if cp in _DEC else cp & 0xFF)  # src/gpt.cc:92-96
    return bytes(out)


# ------------------------------------------------------------------------------------------ regexp
def _c_locale_space(pattern: str) -> str:
    """\\s / \\S spelled out as the C-locale set PCRE2 uses without PCRE2_UCP (space, \\t..\\r)."""
    out, i, in_class = [], 0, False
    while i < len(pattern):
        c = pattern[i]
        if c == "\\" and i + 1 < len(pattern):
            n = pattern[i + 1]
            if n == "s":
                out.append(" \\t-\\r" if in_class else "[ \\t-\\r]")
            elif n == "S":
                if in_class:
                    raise ValueError("\\S inside a class is not rewritten")
                out.append("[^ \\t-\\r]")
            else:
                out.append(c + n)
            i += 2
            continue
        if c == "[" and not in_class:
            in_class = True
        elif c == "]" and in_class:
            in_class = False
        out.append(c)
        i += 1
    return "".join(out)


def split(pattern: str, subject: bytes) -> list[bytes]:
    """The pieces regexp_iterator yields.  PCRE2 without PCRE2_UTF reads the subject as code points
    0..255 (\\p{L} sees Latin-1 letters), and without PCRE2_UCP \\s is the C-locale set: the same
    thing is a str of Latin-1 characters matched by a Unicode engine with \\s spelled out."""
    rx = regex.compile(_c_locale_space(pattern), regex.V0)
    text = subject.decode("latin-1")
    pieces, offset = [], 0
    while True:
        m = rx.search(text, offset)
        if m is None:
            break
        length = m.end() - m.start()
        if m.end() == offset:
            raise RuntimeError("regexp_iterator: empty match")
        pieces.append(subject[offset:offset + length])  # src/regexp.cc:146-155: cut at the previous end
        offset = m.end()
        if offset == len(text):
            break
    return pieces


# ------------------------------------------------------------------------------------------ byte_pair_encoder
class Tokenizer:
    def __init__(self, pattern: str = LLAMA3_PATTERN):
        self.pattern = pattern
        self.forward: dict[bytes, int] = {}
        self.inverse: dict[int, bytes] = {}
        self.control: dict[int, int] = {}

    def insert(self, value: bytes, key: int, kind: int = REGULAR):
        self.forward[value] = key
        self.inverse[key] = value
        if kind != REGULAR:
            self.control[kind] = key

    def insert_back(self, value: bytes, kind: int = REGULAR):
        self.insert(value, len(self.forward), kind)

    def insert_control_tokens(self):
        res = lambda i: b"<|reserved_special_token_%d|>" % i
        for value, kind in ((b"<|begin_of_text|>", BEGIN_TEXT), (b"<|end_of_text|>", END_TEXT), (res(0), RESERVED),
                            (res(1), RESERVED), (b"<|finetune_right_pad_id|>", FINETUNE_RIGHT_PAD), (res(2), RESERVED),
                            (b"<|start_header_id|>", BEGIN_HEADER), (b"<|end_header_id|>", END_HEADER),
                            (b"<|eom_id|>", END_MESSAGE), (b"<|eot_id|>", END_TURN), (b"<|python_tag|>", IPYTHON)):
            self.insert_back(value, kind)

    @classmethod
    def from_tiktoken_lines(cls, lines, pattern: str = LLAMA3_PATTERN):
        t = cls(pattern)
        for line in lines:
            key_part, _, value_part = line.partition(" ")
            t.insert(base64.b64decode(key_part), int(value_part))
        t.insert_control_tokens()
        return t

    def _rank(self, key: bytes) -> int:
        return self.forward.get(key, LIMIT)

    def merge_piece(self, s: bytes) -> list[int]:
        """bpe.h:120-168.  Segment i starts at byte i; every byte but the LAST gets one, the last slot
        is the end marker (rank LIMIT).  Segments are visited in (rank, start) order -- stale queue
        entries included -- and a visited live segment swallows its right neighbour when the
        concatenation is a token."""
        n = len(s)
        seg = [[self._rank(s[i:i + 1]), i + 1] for i in range(n - 1)] + [[LIMIT, n]]
        heap = [(seg[i][0], i) for i in range(n - 1)]
        heapq.heapify(heap)
        while heap:
            _, begin = heapq.heappop(heap)
            nxt = seg[begin][1]
            if seg[begin][0] >= LIMIT or nxt >= len(seg):
                continue
            end = seg[nxt][1]
            merged = self._rank(s[begin:end])
            if merged >= LIMIT:
                continue
            heapq.heappush(heap, (merged, begin))
            seg[begin] = [merged, end]
            seg[nxt][0] = LIMIT
        return [r for r, _ in seg if r < LIMIT]

    def encode(self, text: bytes) -> list[int]:
        out = []
        for piece in split(self.pattern, text):
            if piece in self.forward:
                out.append(self.forward[piece])
            else:
                out.extend(self.merge_piece(piece))
        return out

    def encode_control(self, kind: int) -> int:
        if kind not in self.control:
            raise KeyError(f"byte_pair_encoder: unknown control token '{kind}'")
        return self.control[kind]

    def decode(self, ids) -> bytes:
        return b"".join(self.inverse[i] for i in ids)


# ------------------------------------------------------------------------------------------ scanners
class SentencePiece(Tokenizer):
    """text::sentence_piece (include/metalchat/text/sentence_piece.h:17-104): byte_pair_encoder<char32_t> with the token
    pattern ".*", spaces written as U+2581.  Keys are `str` here -- merge_piece slices code points, as the reference's
    std::u32string does.  The reference's ".*" (PCRE2, no DOTALL) stops at a line feed and its iterator then does not
    advance (src/regexp.cc:146-160): a text with a line feed never comes back.  Lines are pieces here and each line feed is
    a piece of its own -- the one place where this restatement cannot follow the reference."""

    SPACE, MARK = " ", "\u2581"

    def __init__(self):
        super().__init__(".*")

    @classmethod
    def from_hf_json(cls, doc: dict):
        """huggingface::gemma3_tokenizer_loader::load (src/gemma.cc:72-94)"""
        t = cls()
        for value, key in doc["model"]["vocab"].items():
            t.insert(value, key)
        for tok in doc.get("added_tokens", []):
            t.insert(tok["content"], tok["id"], tok["id"])  # text::tokenkind(token.id)
        return t

    def encode(self, text: str) -> list[int]:
        out = []
        s = text.replace(self.SPACE, self.MARK)
        pieces, b = [], 0
        for i, ch in enumerate(s):
            if ch == "\n":
                pieces += [s[b:i], s[i:i + 1]]
                b = i + 1
        pieces.append(s[b:])
        for piece in pieces:
            if not piece:
                continue
            if piece in self.forward:
                out.append(self.forward[piece])
            else:
                out.extend(self.merge_piece(piece))
        return out

    def decode(self, ids) -> str:
        return "".join(self.inverse[i].replace(self.MARK, self.SPACE) for i in ids)


class LimitScanner:
    def __init__(self, lim):
        self.lim, self.n = lim, 0

    def reset(self):
        self.n = 0

    def scan(self, _tok):
        self.n += 1
        return self.n < self.lim


class MatchScanner:
    def __init__(self, tokens):
        self.tokens = set(tokens)

    def reset(self):
        pass

    def scan(self, tok):
        return tok not in self.tokens


class CompositeScanner:
    def __init__(self, parts, op_and=True):
        self.parts, self.op_and = list(parts), op_and

    def reset(self):
        for p in self.parts:
            p.reset()

    def scan(self, tok):
        if not self.parts:
            return False
        r = self.parts[0].scan(tok)
        for p in self.parts[1:]:
            v = p.scan(tok)
            r = (r and v) if self.op_and else (r or v)
        return r


# ------------------------------------------------------------------------------------------ interpreter framing
def header_ids(tok: Tokenizer, role: bytes) -> list[int]:
    return [tok.encode_control(BEGIN_HEADER)] + tok.encode(role) + [tok.encode_control(END_HEADER)] + tok.encode(b"\n\n")


def message_ids(tok: Tokenizer, role: bytes, content: bytes) -> list[int]:
    return header_ids(tok, role) + tok.encode(content) + [tok.encode_control(END_TURN)]


def read_until(step, scanner, first_token: int):
    """interpreter.h:358-374 with `step(token) -> next token`: the ids that are decoded."""
    scanner.reset()
    out, token = [], first_token
    while scanner.scan(token):
        out.append(token)
        token = step(token)
    return out
