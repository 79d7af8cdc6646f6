"""SURVEY.md s.8f-4 on the CPU: GPT-2 codec, split pattern, byte-pair encoder, llama3 loaders, scanners
and message framing of the C ABI (Part 4) against the reference's known answers
(test/test_bpe.cc) and the restatement in oracle/text_oracle.py.

Pinned by the reference's own tests: the GPT-2 codec vectors, the control-token ids.  The encoder's
sentence vectors need Meta's tokenizer.model (not in the reference tree): they run when
MC_LLAMA3_TOKENIZER_MODEL points at it, otherwise the encoder is checked against the restatement and
the split against an independent regex engine -- parity unpinned, see the oracle's header."""
import base64
import collections
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import metalchat_amd as mc  # noqa: E402
import text_oracle as to  # noqa: E402

CORPUS = ("This is a test sentence. And his name is John Cena. This is debatable topic.\n"
          "The quick brown fox jumps over the lazy dog, doesn't it? I'll say: we've 12345 reasons & 67 more!\n\n"
          "  indented    text\twith\ttabs \r\n and trailing spaces   \n"
          "def multiply(a, b):\n    return a * b  # 113001120 == 12135 * 9312\n"
          "ipython Environment: tools, JSON {\"name\":\"multiply\",\"parameters\":{\"a\":\"12135\"}} "
          "café naïve استاندارد 中文 \U0001F600 done")


def train_vocab(corpus: bytes, n_merges: int):
    """A small byte-level BPE vocabulary (256 bytes + learned merges, rank = id) so that pieces
    merge over several levels, as with a real token map."""
    pieces = to.split(to.LLAMA3_PATTERN, corpus)
    words = collections.Counter(tuple(bytes([b]) for b in p) for p in pieces)
    vocab = [bytes([b]) for b in range(256)]
    for _ in range(n_merges):
        pairs = collections.Counter()
        for w, c in words.items():
            for a, b in zip(w, w[1:]):
                pairs[(a, b)] += c
        if not pairs:
            break
        (a, b), _ = max(pairs.items(), key=lambda kv: (kv[1], kv[0]))
        vocab.append(a + b)
        nw = collections.Counter()
        for w, c in words.items():
            out, i = [], 0
            while i < len(w):
                if i + 1 < len(w) and w[i] == a and w[i + 1] == b:
                    out.append(a + b)
                    i += 2
                else:
                    out.append(w[i])
                    i += 1
            nw[tuple(out)] += c
        words = nw
    return vocab


@pytest.fixture(scope="module")
def vocab():
    v = train_vocab(CORPUS.encode("utf-8") * 2, 300)
    assert len(set(v)) == len(v)
    return v


@pytest.fixture(scope="module")
def tiktoken_file(vocab, tmp_path_factory):
    # shuffled line order: the rank comes from the line, not from its position
    p = tmp_path_factory.mktemp("tok") / "tokenizer.model"
    order = np.random.default_rng(3).permutation(len(vocab))
    with open(p, "w") as f:
        for i in order:
            f.write(base64.b64encode(vocab[i]).decode() + " " + str(int(i)) + "\n")
    return str(p)


@pytest.fixture(scope="module")
def pair(vocab, tiktoken_file):
    t = mc.Tokenizer.open_tiktoken(tiktoken_file)
    o = to.Tokenizer.from_tiktoken_lines(base64.b64encode(v).decode() + " " + str(i) for i, v in enumerate(vocab))
    yield t, o
    t.release()


# ------------------------------------------------------------------------------------------ gpt2 codec
def test_gpt2_codec_reference_known_answers():
    # test/test_bpe.cc:29-36
    assert mc.gpt2_encode(b"    Hello  \x80") == "ĠĠĠĠHelloĠĠĢ"
    assert mc.gpt2_decode("ĠĠĠĠHelloĠĠĢ") == b"    Hello  \x80"
    # test/test_bpe.cc:47-54 (the string the reference decodes from id 125579)
    s = " استاندارد"
    coded = mc.gpt2_encode(s.encode("utf-8"))
    assert coded == "ĠØ§Ø³ØªØ§ÙĨØ¯Ø§Ø±Ø¯"
    assert mc.gpt2_decode(coded).decode("utf-8") == s


def test_gpt2_codec_is_gpt2s_published_byte_alphabet():
    # the table GPT-2's encoder.py builds (bytes_to_unicode), written out independently
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAC + 1)) + list(range(0xAE, 0xFF + 1))
    cs, n = bs[:], 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    table = dict(zip(bs, cs))
    every = bytes(range(256))
    assert mc.gpt2_encode(every) == "".join(chr(table[b]) for b in every) == to.gpt2_encode(every)
    assert mc.gpt2_decode(mc.gpt2_encode(every)) == every


def test_gpt2_decode_keeps_the_low_byte_of_foreign_code_points_and_rejects_bad_utf8():
    # src/gpt.cc:92-96: a code point outside the table falls through as char(rune)
    assert mc.gpt2_decode("中") == b"\x2d" == to.gpt2_decode("中")
    assert mc.gpt2_decode(" ") == b" "  # U+0020 is not in the table either
    with pytest.raises(mc.McError, match="UTF-8"):
        mc.runtime._bytes_call(mc.capi().mc_gpt2_decode, b"\xff\xfe", 2)
    with pytest.raises(mc.McError, match="U\\+FFFF"):
        mc.gpt2_decode("\U0001F600")


# ------------------------------------------------------------------------------------------ split
def test_split_matches_an_independent_engine_on_the_corpus_and_on_noise():
    rng = np.random.default_rng(11)
    subjects = [CORPUS.encode("utf-8"), b"", b" ", b"\n", b"a", b"'S'T'Re 'LL", b"   x", b"x   ", b"\r\n\r\n  \t"]
    alphabet = b" \t\n\r'.,!?abcXYZ019_-/{}\":" + bytes([0x85, 0xA0, 0xB2, 0xB5, 0xC3, 0xA9, 0xD8, 0xFF, 0x1C, 0x00])
    for _ in range(300):
        n = int(rng.integers(1, 40))
        subjects.append(bytes(alphabet[i] for i in rng.integers(0, len(alphabet), n)))
    for _ in range(100):
        subjects.append(rng.integers(0, 256, int(rng.integers(1, 48)), dtype=np.uint8).tobytes())
    for s in subjects:
        got, want = mc.regexp_split(None, s), to.split(to.LLAMA3_PATTERN, s)
        assert got == want, s
        assert b"".join(got) == s  # this pattern leaves no byte unmatched


def test_split_reads_bytes_not_utf8():
    # src/regexp.cc:38-41: compiled with no options, so 0xC3 is the Latin-1 letter and 0xA9 the symbol
    assert mc.regexp_split(None, " café!".encode("utf-8")) == [b" caf\xc3", b"\xa9!"]


def test_split_cuts_at_the_previous_end_with_the_new_length():
    # src/regexp.cc:146-155 with a pattern that skips text: the piece is NOT the matched text
    assert mc.regexp_split(b"[0-9]+", b"ab12cd345") == to.split("[0-9]+", b"ab12cd345") == [b"ab", b"cd3"]


def test_split_errors():
    with pytest.raises(mc.McError, match="regexp: invalid regular expression"):
        mc.regexp_split(b"(unclosed", b"x")
    with pytest.raises(mc.McError, match="empty match"):
        mc.regexp_split(b"x*", b"aaa")


# ------------------------------------------------------------------------------------------ encoder
def test_encode_decode_follow_the_restatement(pair):
    t, o = pair
    rng = np.random.default_rng(5)
    texts = [CORPUS.encode("utf-8")] + [l.encode("utf-8") for l in CORPUS.split("\n")]
    words = CORPUS.split()
    for _ in range(200):
        k = int(rng.integers(1, 12))
        texts.append(" ".join(words[i] for i in rng.integers(0, len(words), k)).encode("utf-8"))
    for _ in range(100):
        texts.append(rng.integers(0, 256, int(rng.integers(1, 64)), dtype=np.uint8).tobytes())
    for s in texts:
        ids = t.encode(s)
        assert ids == o.encode(s), s
        assert t.decode(ids) == o.decode(ids)
    # a piece that IS a token is looked up whole (bpe.h:288-291)
    assert t.encode(b" the") == [o.forward[b" the"]]
    # ASCII text whose pieces merge completely survives the round trip
    s = b"This is a test sentence."
    assert t.decode(t.encode(s)) == s


def test_merge_drops_an_unmerged_last_byte():
    # bpe.h:137-145: the loop stops one byte short and the end marker carries no token
    t = mc.Tokenizer.create()
    o = to.Tokenizer()
    for i, v in enumerate([b"a", b"b", b"c", b"ab"]):
        t.insert(v, i)
        o.insert(v, i)
    for s, want in ((b"ab", [3]), (b"ac", [0]), (b"abc", [3]), (b"cab", [2, 3]), (b"b", [1]), (b"z", []), (b"zz", []),
                    (b"abab", [3, 3]), (b"aba", [3]), (b"abz", [3])):
        assert t.encode(s) == o.encode(s) == want, s
    t.release()


def test_segments_are_visited_by_their_own_rank_not_by_pair_rank():
    # where the reference parts with tiktoken: "abc" with tokens bc < ab.  tiktoken merges the
    # lower-ranked PAIR first (a, bc); the reference visits segment a (rank 0) first and joins ab.
    t = mc.Tokenizer.create()
    o = to.Tokenizer()
    for i, v in enumerate([b"a", b"b", b"c", b"d", b"bc", b"ab"]):
        t.insert(v, i)
        o.insert(v, i)
    assert t.encode(b"abcd") == o.encode(b"abcd") == [5, 2]
    t.release()


def test_control_tokens_sit_behind_the_map_at_llama3s_ids(tmp_path):
    # reference::llama3_tokenizer_loader::insert_control_tokens (src/reference.cc:113-127) behind a
    # 128000-entry token map: the published Llama-3 ids; test/test_bpe.cc:122-132 decodes 128001
    p = tmp_path / "big.model"
    with open(p, "w") as f:
        for i in range(128000):
            f.write(base64.b64encode(b"t%d" % i).decode() + " %d\n" % i)
    t = mc.Tokenizer.open_tiktoken(str(p))
    assert len(t) == 128011
    assert t.decode(128001) == b"<|end_of_text|>"
    want = {mc.TOKEN_BEGIN_TEXT: 128000, mc.TOKEN_END_TEXT: 128001, mc.TOKEN_FINETUNE_RIGHT_PAD: 128004,
            mc.TOKEN_RESERVED: 128005, mc.TOKEN_BEGIN_HEADER: 128006, mc.TOKEN_END_HEADER: 128007,
            mc.TOKEN_END_MESSAGE: 128008, mc.TOKEN_END_TURN: 128009, mc.TOKEN_IPYTHON: 128010}
    for kind, key in want.items():
        assert t.encode_control(kind) == key
    assert t.decode([128002, 128003, 128005]) == (b"<|reserved_special_token_0|><|reserved_special_token_1|>"
                                                   b"<|reserved_special_token_2|>")
    assert t.decode([128006, 128007, 128009, 128010]) == b"<|start_header_id|><|end_header_id|><|eot_id|><|python_tag|>"
    with pytest.raises(mc.McError, match="byte_pair_encoder: unknown control token '1'"):
        t.encode_control(mc.TOKEN_REGULAR)
    with pytest.raises(mc.McError, match="byte_pair_encoder: unable to decode id '128011'"):
        t.decode(128011)
    t.release()


def test_huggingface_tokenizer_json_gives_the_same_encoder(vocab, pair, tmp_path):
    t, o = pair
    doc = {"version": "1.0",
           "pre_tokenizer": {"type": "Sequence", "pretokenizers": [
               {"type": "Split", "pattern": {"Regex": to.LLAMA3_PATTERN}, "behavior": "Isolated", "invert": False},
               {"type": "ByteLevel", "add_prefix_space": False, "trim_offsets": True, "use_regex": False}]},
           "model": {"type": "BPE", "vocab": {to.gpt2_encode(v): i for i, v in enumerate(vocab)}, "merges": []}}
    p = tmp_path / "tokenizer.json"
    p.write_text(json.dumps(doc, ensure_ascii=False), encoding="utf-8")
    h = mc.Tokenizer.open_hf(str(p))
    assert len(h) == len(t)
    for line in CORPUS.split("\n"):
        assert h.encode(line) == t.encode(line)
    assert h.encode_control(mc.TOKEN_END_TURN) == t.encode_control(mc.TOKEN_END_TURN)
    h.release()
    # src/llama.cc:97-102
    doc["pre_tokenizer"] = {"type": "ByteLevel"}
    p.write_text(json.dumps(doc))
    with pytest.raises(mc.McError, match="does not provide an input sequence regular expression"):
        mc.Tokenizer.open_hf(str(p))
    with pytest.raises(mc.McError, match="llama3_tokenizer_loader: failed opening file"):
        mc.Tokenizer.open_hf(str(tmp_path / "absent.json"))
    with pytest.raises(mc.McError, match="llama3_tokenizer_loader: failed opening file"):
        mc.Tokenizer.open_tiktoken(str(tmp_path / "absent.model"))


@pytest.mark.skipif(not os.environ.get("MC_LLAMA3_TOKENIZER_MODEL"), reason="needs Meta's Llama-3 tokenizer.model")
def test_reference_known_answers_with_the_real_token_map():
    # test/test_bpe.cc:58-132
    t = mc.Tokenizer.open_tiktoken(os.environ["MC_LLAMA3_TOKENIZER_MODEL"])
    assert t.encode("This is a test sentence.") == [2028, 374, 264, 1296, 11914, 13]
    assert t.decode([2028, 374, 264, 1296, 11914, 13]) == b"This is a test sentence."
    assert t.encode("And his name is John Cena.") == [3112, 813, 836, 374, 3842, 89663, 13]
    assert t.decode(t.encode(" ipython")) == b" ipython"
    assert len(t.encode("This is debatable topic.")) > 0
    assert t.decode(125579).decode("utf-8") == " استاندارد"
    assert t.decode(128001) == b"<|end_of_text|>"
    t.release()


# ------------------------------------------------------------------------------------------ interpreter framing
def test_message_framing_and_variables(pair):
    t, o = pair
    it = mc.Interpreter(None, t)
    bot = o.encode_control(to.BEGIN_TEXT)
    assert it.pending() == [bot]  # src/interpreter.cc:79-80
    it.declare_variable("extra_instructions", "answer in json")
    it.write("system", "You are a test. {{ extra_instructions }}{{unknown}}!")
    it.write("user", "What is 12135 multiplied by 9312?")
    want = [bot] + to.message_ids(o, b"system", b"You are a test. answer in json!") + \
        to.message_ids(o, b"user", b"What is 12135 multiplied by 9312?")
    assert it.pending() == want
    assert it.start_pos == 0
    with pytest.raises(mc.McError, match="without a decoder"):
        it.read()
    it.release()


# ------------------------------------------------------------------------------------------ text::sentence_piece (gemma3)
def _sp_vocab():
    """A small sentence-piece style vocabulary: every code point of the corpus, then merges learnt greedily over code
    points (ids = merge order, as a published tokenizer.json has them)."""
    corpus = "the quick brown fox jumps over the lazy dog  and the d\u00e9j\u00e0 vu caf\u00e9 \u00fcber na\u00efve \u4f60\u597d \u4e16\u754c hello world the end\nsecond line here"
    text = corpus.replace(" ", "\u2581")
    vocab = {}
    for special in ("<pad>", "<eos>", "<bos>", "<unk>"):
        vocab[special] = len(vocab)
    for ch in sorted(set(text)):
        vocab[ch] = len(vocab)
    words = [list(w) for w in text.replace("\n", " \n ").split(" ") if w]
    for _ in range(60):
        pairs = collections.Counter()
        for w in words:
            for a, b in zip(w, w[1:]):
                pairs[a + b] += 1
        pairs = {k: v for k, v in pairs.items() if k not in vocab}
        if not pairs:
            break
        best = max(sorted(pairs), key=lambda k: pairs[k])
        vocab[best] = len(vocab)
        for w in words:
            i = 0
            while i + 1 < len(w):
                if w[i] + w[i + 1] == best:
                    w[i:i + 2] = [best]
                else:
                    i += 1
    return vocab


@pytest.fixture(scope="module")
def sp_pair(tmp_path_factory):
    vocab = _sp_vocab()
    doc = {"model": {"type": "BPE", "vocab": vocab},
           "added_tokens": [{"id": 1, "content": "<eos>"}, {"id": 2, "content": "<bos>"},
                            {"id": len(vocab), "content": "<start_of_turn>"}, {"id": len(vocab) + 1, "content": "<end_of_turn>"}]}
    path = tmp_path_factory.mktemp("sp") / "tokenizer.json"
    path.write_text(json.dumps(doc, ensure_ascii=False), encoding="utf-8")
    t = mc.Tokenizer.open_hf_gemma3(str(path))
    yield t, to.SentencePiece.from_hf_json(doc), doc
    t.release()


def test_sentence_piece_follows_the_restatement(sp_pair):
    t, o, doc = sp_pair
    assert len(t) == len(o.forward)
    rng = np.random.default_rng(5)
    alphabet = list("the quickbrownfxjmpsvlazydg \u00e9\u00e0\u00fc\u00ef\u4f60\u597d\u4e16\u754c") + ["\n", "  ", " the", "caf\u00e9", "\U0001F600", "Q"]
    texts = ["the quick brown fox", " hello world ", "d\u00e9j\u00e0 vu", "\u4f60\u597d \u4e16\u754c", "the end\nsecond line", "x", "", "\n\n", "\u2581already"]
    texts += ["".join(rng.choice(alphabet, size=int(rng.integers(1, 40)))) for _ in range(300)]
    for text in texts:
        ids = t.encode(text)
        assert ids == o.encode(text), text
        assert t.decode(ids).decode("utf-8") == o.decode(ids)


def test_sentence_piece_spaces_lines_and_the_dropped_last_code_point(sp_pair):
    t, o, doc = sp_pair
    vocab = doc["model"]["vocab"]
    # a piece that is a token goes through whole; spaces are U+2581 on the way in and spaces again on the way out
    ids = t.encode("the")
    assert ids == [vocab["the"]] if "the" in vocab else True
    assert " " not in "".join(k for k in vocab)
    s = "the quick brown fox jumps over the lazy dog"
    assert t.decode(t.encode(s)).decode("utf-8") in (s, s[:-1])  # bpe.h:120-168: an unmerged LAST unit is dropped
    # units are code points: a two-byte letter that is a token of its own is never cut into bytes
    e = t.encode("\u00e9\u00e9\u00e9")
    assert all(len(o.inverse[i].encode("utf-8")) >= 2 for i in e)
    # a line feed is a piece of its own (the reference does not return from such a text: sentence_piece.h + regexp.cc:146-160)
    assert t.encode("the\nthe") == o.encode("the\nthe")
    # unknown code points vanish like in the reference (rank LIMIT never emitted)
    assert t.encode("\U0001F600") == []


def test_gemma3_loader_binds_added_tokens_to_their_own_id_and_fails_like_the_reference(sp_pair, tmp_path):
    t, o, doc = sp_pair
    for tok in doc["added_tokens"]:
        if tok["id"] == mc.TOKEN_REGULAR:
            # text::tokenkind(token.id) with id 1 IS token::regular (tokenizer.h:29): bound to no kind, as in the reference
            with pytest.raises(mc.McError, match="unknown control token '1'"):
                t.encode_control(tok["id"])
        else:
            assert t.encode_control(tok["id"]) == tok["id"]
        assert t.decode([tok["id"]]).decode() == tok["content"]
    with pytest.raises(mc.McError, match="gemma3_tokenizer_loader: failed opening file"):
        mc.Tokenizer.open_hf_gemma3(str(tmp_path / "absent.json"))
    empty = mc.Tokenizer.create_sentence_piece()
    empty.insert_back("a".encode())
    empty.insert_back("b".encode())
    empty.insert_back("ab".encode())
    assert empty.encode("abab") == [2, 2]
    assert empty.encode("aba") == [2]  # the last unit only survives inside a merge -- bpe.h:120-168
    empty.release()
