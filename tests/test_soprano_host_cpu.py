"""CPU: host-side text front end of the Soprano mirror (Soprano.swift:365-449, 594-624)."""
from mlx_audio_swift_amd import soprano as sp


def test_sentence_split_and_merge():
    assert sp.split_into_sentences("One. Two!  Three? four") == ["One.", "Two!", "Three?", "four"]
    assert sp.split_into_sentences("no boundary") == ["no boundary"]
    long = "This sentence is definitely longer than thirty characters."
    out = sp.preprocess_text(["Hi. " + long + " Ok."])
    # "Hi." (short, first) merges into the next; "Ok." (short) merges into the previous
    assert out == [(f"[STOP][TEXT]Hi. {long} Ok.[START]", 0, 0)]
    out = sp.preprocess_text([long + " " + long, "Short."])
    assert [o[1:] for o in out] == [(0, 0), (0, 1), (1, 0)] and out[2][0] == "[STOP][TEXT]Short.[START]"


def test_split_prompt_chunks():
    assert sp.split_prompt("a\\nb\n\n c ") == ["a", "b", "c"]
    text = ("word " * 30 + "end. ") * 5                     # 775 characters, boundaries every 155
    chunks = sp.split_prompt(text)
    assert len(chunks) == 5 and all(c.endswith("end.") for c in chunks)
    assert sp.split_prompt("x" * 1200) == ["x" * 500, "x" * 500, "x" * 200]
