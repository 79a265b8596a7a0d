"""Sequence-bias table for decoding (reference utils/generation_helper.py:18-73): every word -- and, with the `yake`
keyword extractor, every key phrase -- of the TRAINING sentences becomes a token tuple with a constant bias that
`model.generate(sequence_bias=...)` applies through ns_logits_process (HF SequenceBiasLogitsProcessor semantics).

Same class name, constructor arguments and methods as the reference.  `yake` is an optional dependency here: the
'word' table needs only the tokenizer; 'phrase' / 'phrase_word' raise a clear ImportError without it."""
from utils.reader import read_jsonlines


class GetSequenceBias(object):
    def __init__(self, tokenizer_name, jsonl_path, bias=None, extract_type=None, tokenizer=None):
        self.jsonl = read_jsonlines(jsonl_path)
        self.sentences = [line["sentence"] for line in self.jsonl]
        if tokenizer is None:       # reference :26: the model's tokenizer, words tokenised with a leading space
            from transformers import AutoTokenizer
            tokenizer = AutoTokenizer.from_pretrained(tokenizer_name, add_prefix_space=True)
        self.tokenizer_with_prefix_space = tokenizer
        self._kw = None
        self.bias = self.get_bias_for_sentences(self.sentences, bias, extract_type)
        assert self.bias != {}

    @property
    def kw_extractor(self):
        if self._kw is None:
            try:
                import yake
            except ImportError as e:
                raise ImportError("phrase extraction needs the `yake` package (reference utils/generation_helper.py:27); "
                                  "extract_type='word' works without it") from e
            self._kw = yake.KeywordExtractor(lan="en", n=3, dedupLim=0.9, top=20, features=None)
        return self._kw

    def get_phrases_from_sentence(self, sentence, cannot_be_single_word=False):
        phrases = [p[0] for p in self.kw_extractor.extract_keywords(sentence)]
        return [p for p in phrases if len(p.split()) != 1] if cannot_be_single_word else phrases

    def get_phrases_from_sentences(self, sentences, cannot_be_single_word):
        phrases = []
        for sentence in set(sentences):
            phrases.extend(self.get_phrases_from_sentence(sentence, cannot_be_single_word))
        return phrases

    def get_tokens_as_tuple(self, word):
        tok = self.tokenizer_with_prefix_space
        if hasattr(tok, "encode_text"):     # the synthetic tokenizer of this build: text ids without the prompt / EOS frame
            return tuple(tok.encode_text(word)[4:-1])
        return tuple(tok([word], add_special_tokens=False).input_ids[0])

    def get_tokens_as_tuple_from_sentences(self, sentences):
        words = {word for sentence in sentences for word in sentence.split()}
        return {self.get_tokens_as_tuple(word) for word in words}

    def get_bias_for_tokens(self, tokens, bias):
        return {token: bias for token in tokens}

    def get_bias_for_sentences(self, sentences, bias, extract_type=None):
        if extract_type == "word":
            tokens = self.get_tokens_as_tuple_from_sentences(sentences)
        elif extract_type == "phrase":
            tokens = {self.get_tokens_as_tuple(w) for w in self.get_phrases_from_sentences(sentences, cannot_be_single_word=True)}
        elif extract_type == "phrase_word":
            tokens = {self.get_tokens_as_tuple(w) for w in self.get_phrases_from_sentences(sentences, cannot_be_single_word=False)}
        else:
            raise NotImplementedError
        return {token: bias for token in tokens if len(token) > 0}

    def get_bias_for_my_sentences(self):
        return self.bias
