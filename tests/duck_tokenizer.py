"""A deterministic stand-in for the Llama-2 tokenizer (no tokenizer files exist offline), shared by tests/golden/make_golden_batch_transform.py (which
feeds it to the REFERENCE's RLDSBatchTransform) and tests/test_data_path.py (which feeds it to the mirror).  It reproduces the three properties of the
real tokenizer the batch transform relies on (prismatic/vla/datasets/datasets.py:36-97, action_tokenizer.py:38-47, modeling_prismatic.py:974):
  * the last 256 ids below vocab_size (31744 .. 31999) are single characters, so decode / re-encode of an action string is one token per action value;
  * the space after "Out:" in front of a non-space character becomes the stand-alone token 29871;
  * "</s>" is the stop token 2 and a BOS (1) is prepended with add_special_tokens.
Everything else is tokenised per whitespace-separated word by a hash (the values are irrelevant: they are labelled IGNORE)."""
import zlib

ACTION0, VOCAB, EMPTY, BOS, EOS = 31744, 32000, 29871, 1, 2
PUA = 0xE000      # private-use code points stand for the 256 action tokens


class DuckTokenizer:
    vocab_size = VOCAB

    def decode(self, ids):
        return "".join(chr(PUA + int(i) - ACTION0) for i in ids)

    def batch_decode(self, rows):
        return [self.decode(r) for r in rows]

    def encode_text(self, text: str):
        ids, i, n = [BOS], 0, len(text)
        while i < n:
            c = text[i]
            if text.startswith("</s>", i):
                ids.append(EOS); i += 4
            elif PUA <= ord(c) < PUA + 256:
                ids.append(ACTION0 + ord(c) - PUA); i += 1
            elif c.isspace():
                if c == " " and i + 1 < n and PUA <= ord(text[i + 1]) < PUA + 256:
                    ids.append(EMPTY)          # sentencepiece's lone word-boundary piece in front of a non-space-prefixed token
                i += 1
            else:
                j = i
                while j < n and not text[j].isspace() and not text.startswith("</s>", j) and not (PUA <= ord(text[j]) < PUA + 256):
                    j += 1
                ids.append(3 + zlib.crc32(text[i:j].encode()) % 29000)
                i = j
        return ids

    def __call__(self, text, add_special_tokens=True):
        class Out:
            pass

        o = Out()
        o.input_ids = self.encode_text(text) if add_special_tokens else self.encode_text(text)[1:]
        return o


def mirror_tokenizer(text: str):
    """The callable form the mirror's RLDSBatchTransform / PrismaticProcessor take: text -> ids incl. BOS."""
    return DuckTokenizer().encode_text(text)
