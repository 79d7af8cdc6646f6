"""metalchat::interpreter over the HIP decoder: write / read through the C ABI
against the same conversation driven by hand -- the restated framing (oracle/text_oracle.py), the
restated read_until loop and scanners, and a second decoder with the same weights stepped through
mc_decoder_prefill / mc_decoder_step.  Token ids are compared bit for bit (both sides run the same
kernels on the same inputs; what is under test is the loop: what is flushed, where start_pos goes,
which token stops it and whether it is emitted).  PARITY UNPINNED by the reference: its interpreter
test needs the Llama-3.2-1B checkpoint (test/test_interpreter.cc:38-84)."""
import base64
import os
import sys

import numpy as np
import pytest

import modelgen as mg

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import text_oracle as to  # noqa: E402

pytestmark = pytest.mark.gpu
F32 = 1


@pytest.fixture(scope="module")
def toks(tmp_path_factory):
    import metalchat_amd as mc

    # 256 bytes + a few words: enough to frame messages; ids stay below the model's vocabulary
    vocab = [bytes([b]) for b in range(256)] + [b"as", b"sis", b"assis", b"tant", b"assistant", b"us", b"er", b"user",
                                                b"\n\n", b" the", b"sy", b"stem", b"system", b"He", b"llo", b"Hello"]
    vocab += [b"<w%d>" % i for i in range(501 - len(vocab))]   # 501 + 11 control tokens = the model's 512 ids
    p = tmp_path_factory.mktemp("itok") / "tokenizer.model"
    with open(p, "w") as f:
        for i, v in enumerate(vocab):
            f.write(base64.b64encode(v).decode() + " %d\n" % i)
    t = mc.Tokenizer.open_tiktoken(str(p))
    o = to.Tokenizer.from_tiktoken_lines(base64.b64encode(v).decode() + " %d" % i for i, v in enumerate(vocab))
    yield t, o
    t.release()


def make_decoders(acc, n=2, **over):
    import metalchat_amd as mc

    cfg = mg.tiny_cfg(F32, vocab=512, max_seq_len=160, **over)
    weights = mg.make_model(cfg, seed=5)
    decs = []
    for _ in range(n):
        d = mc.Decoder(acc, **mg.decoder_kwargs(cfg))
        d.load_model(weights)
        decs.append(d)
    return cfg, decs


def by_hand(dec, pending, scanner, start_pos):
    """interpreter::read by hand -> (ids emitted, start_pos afterwards)"""
    token = dec.prefill(pending, start_pos) if len(pending) > 1 else dec.step(pending[0], start_pos)
    pos = [start_pos + len(pending)]

    def step(tok):
        nxt = dec.step(tok, pos[0])
        pos[0] += 1
        return nxt

    ids = to.read_until(step, scanner, token)
    return ids, pos[0]


def test_two_turns_match_the_loop_driven_by_hand(acc, toks):
    import metalchat_amd as mc

    t, o = toks
    cfg, (da, db) = make_decoders(acc)
    it = mc.Interpreter(da, t)
    it.set_token_scanner(limit=12, stop_ids=[o.encode_control(to.END_TURN), o.encode_control(to.END_TEXT)], op_and=True)
    scanner = to.CompositeScanner([to.LimitScanner(12), to.MatchScanner([o.encode_control(to.END_TURN),
                                                                          o.encode_control(to.END_TEXT)])], True)
    it.write("system", "Hello the system")
    it.write("user", "Hello")
    pending = [o.encode_control(to.BEGIN_TEXT)] + to.message_ids(o, b"system", b"Hello the system") + \
        to.message_ids(o, b"user", b"Hello") + to.header_ids(o, b"assistant")
    assert it.pending() + to.header_ids(o, b"assistant") == pending
    text, ids = it.read()
    want, pos = by_hand(db, pending, scanner, 0)
    assert ids == want and len(ids) == 11          # limit_token_scanner(12) lets 11 through (interpreter.h:117-121)
    assert text == o.decode(want)
    assert it.start_pos == pos == len(pending) + len(want)
    assert it.pending() == []
    # second turn continues at start_pos with only the new message in the buffer
    it.write("user", "the user")
    pending2 = to.message_ids(o, b"user", b"the user") + to.header_ids(o, b"assistant")
    text2, ids2 = it.read()
    want2, pos2 = by_hand(db, pending2, scanner, pos)
    assert ids2 == want2 and text2 == o.decode(want2)
    assert it.start_pos == pos2
    it.release()
    da.release()
    db.release()


def test_stop_token_ends_the_turn_and_is_not_emitted(acc, toks):
    import metalchat_amd as mc

    t, o = toks
    cfg, (da, db) = make_decoders(acc)
    pending = [o.encode_control(to.BEGIN_TEXT)] + to.message_ids(o, b"user", b"Hello") + to.header_ids(o, b"assistant")
    free, _ = by_hand(db, pending, to.LimitScanner(30), 0)
    assert len(free) == 29
    stop = free[7]                                   # make the 8th generated token a stop token
    first = free.index(stop)
    it = mc.Interpreter(da, t)
    it.set_token_scanner(limit=100, stop_ids=[stop, 511], op_and=True)
    it.write("user", "Hello")
    text, ids = it.read()
    assert ids == free[:first] and text == o.decode(free[:first])
    # flushed prompt + one position per emitted token: the stop token was produced but never fed back
    assert it.start_pos == len(pending) + first
    it.release()
    # std::logical_or: the turn goes on while EITHER scanner says so -- it ends at the first stop token
    # at or past the limit (every scanner sees every token)
    cfg2, (dc,) = make_decoders(acc, n=1)
    stop2 = free[10]
    it = mc.Interpreter(dc, t)
    it.set_token_scanner(limit=5, stop_ids=[stop2], op_and=False)
    it.write("user", "Hello")
    _, ids = it.read()
    replay = iter(free[1:])                          # greedy decoding: the same stream as the free run
    want = to.read_until(lambda tok: next(replay),
                         to.CompositeScanner([to.LimitScanner(5), to.MatchScanner([stop2])], False), free[0])
    assert len(want) == min(i for i in range(4, 11) if free[i] == stop2)
    assert ids == want
    it.release()
    for d in (da, db, dc):
        d.release()


def test_default_scanner_and_empty_composite(acc, toks):
    import metalchat_amd as mc

    t, o = toks
    cfg, (da, db) = make_decoders(acc)
    it = mc.Interpreter(da, t)
    it.write("user", "Hello")
    _, ids = it.read()
    assert len(ids) == 49                            # limit_token_scanner(50), src/interpreter.cc:72
    it.release()
    it = mc.Interpreter(db, t)
    it.set_token_scanner()                           # composite of nothing: scan() is false (interpreter.h:148-152)
    it.write("user", "Hello")
    text, ids = it.read()
    assert ids == [] and text == b""
    assert it.start_pos == 1 + len(to.message_ids(o, b"user", b"Hello")) + len(to.header_ids(o, b"assistant"))
    it.release()
    da.release()
    db.release()


def test_a_chat_longer_than_the_cache_keeps_going_and_a_straddling_turn_is_recoverable(acc, toks):
    # nn::sink_cache::copy (nn/cache.h:167-216): turns whose prompt starts behind the end of the cache rotate the post-sink
    # region by their length and go on; a turn that starts inside the cache and ends outside is an error in the reference
    # (clone's same_numel check) -- here it must leave the pending tokens pending so that the caller can recover.
    import metalchat_amd as mc

    t, o = toks
    cfg = mg.tiny_cfg(F32, vocab=512, max_seq_len=48, n_layers=1)
    weights = mg.make_model(cfg, seed=5)

    def pair():
        ds = []
        for _ in range(2):
            d = mc.Decoder(acc, **mg.decoder_kwargs(cfg))
            d.load_model(weights)
            ds.append(d)
        return ds

    header = to.header_ids(o, b"assistant")
    # (1) a turn that straddles the end of the cache
    da, db = pair()
    it = mc.Interpreter(da, t)
    it.set_token_scanner(limit=6)
    straddled = False
    for turn in range(20):
        it.write("user", "Hello the system")
        pending, start = it.pending(), it.start_pos
        full = pending + header
        if start < 48 < start + len(full):
            with pytest.raises(mc.McError, match="straddle the end of the cache"):
                it.read()
            assert it.start_pos == start and it.pending() == pending   # nothing consumed, nothing lost, no stray header
            straddled = True
            break
        _, ids = it.read()
        want, pos = by_hand(db, full, to.LimitScanner(6), start)
        assert ids == want and it.start_pos == pos, f"turn {turn} at start_pos {start}"
    assert straddled
    it.release(), da.release(), db.release()
    # (2) a conversation whose turns never straddle: it runs far past the cache size (chunks behind a full cache)
    da, db = pair()
    it = mc.Interpreter(da, t)
    it.set_token_scanner(limit=6)
    # the first prompt is sized to end 43 .. 48 rows into the cache, so that the 5 tokens the turn emits carry start_pos
    # to or past the end: every later turn starts behind a full cache
    text = "Hello the system Hello the system Hello the system Hello"
    while True:
        probe = mc.Interpreter(None, t)
        probe.write("user", text)
        n1 = len(probe.pending()) + len(header)
        probe.release()
        if n1 >= 43:
            break
        text += " the"
    assert n1 <= 48
    it.write("user", text)
    done = 0
    while it.start_pos < 200:
        pending, start = it.pending(), it.start_pos
        full = pending + header
        assert not (start < 48 < start + len(full)), "this conversation was built not to straddle the cache end"
        _, ids = it.read()
        want, pos = by_hand(db, full, to.LimitScanner(6), start)
        assert ids == want and it.start_pos == pos, f"turn at start_pos {start}"
        it.write("user", "Hello")
        done += 1
    assert done >= 3 and it.start_pos > 48 * 3
    it.release(), da.release(), db.release()


def test_sentence_piece_interpreter_reads_spaces_not_u2581(acc):
    """interpreter::read_until decodes every emitted token through tokenizer_traits::decode, which for text::sentence_piece
    writes U+2581 back as a space (interpreter.h:365, sentence_piece.h:84-97): the text a gemma3 chat returns has spaces."""
    import metalchat_amd as mc

    cfg, (da,) = make_decoders(acc, n=1)
    t = mc.Tokenizer.create_sentence_piece()
    # every regular id of the model's vocabulary spells "<U+2581>w<i>": whatever the decoder emits carries a space
    n_regular = 512 - 11
    for i in range(n_regular):
        t.insert_back(("▁w%d" % i).encode())
    for kind in (mc.TOKEN_BEGIN_TEXT, mc.TOKEN_END_TEXT, mc.TOKEN_RESERVED, mc.TOKEN_FINETUNE_RIGHT_PAD, mc.TOKEN_BEGIN_HEADER,
                 mc.TOKEN_END_HEADER, mc.TOKEN_END_MESSAGE, mc.TOKEN_END_TURN, mc.TOKEN_IPYTHON):
        t.insert_back(("<|c%d|>" % kind).encode(), kind)
    it = mc.Interpreter(da, t)
    it.set_token_scanner(limit=9)
    it.write("user", " w1 w2")
    text, ids = it.read()
    assert len(ids) == 8
    regular = [i for i in ids if i < n_regular]
    assert regular, ids
    text = text.decode("utf-8") if isinstance(text, bytes) else text
    assert "▁" not in text
    assert text == t.decode(ids).decode("utf-8")
    assert text.count(" w") == len(regular)
    it.release()
    t.release()
    da.release()
