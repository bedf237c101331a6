"""Pure-Python semantic matcher (oracle A', small cases only).  TEST INFRASTRUCTURE ONLY.

An independent statement of the matching rule of SURVEY.md section 8a-0, derived from
TRIE/NTopicFilterTrieNode.java:143-152 ('#' matches the parent level),
TRIE/TopicTrieNode.java:150-152 ('$'-prefixed first level is not wildcard-matchable) and
DW/TopicIndex.java:48-49,55-58.  It deliberately shares no code with bmq_oracle.cpp so the
two can be cross-checked (tests/test_oracle_golden.py).
"""
from typing import List


def parse(topic: str) -> List[str]:
    """UTIL/TopicUtil.java:206-225: split on '/', keep empty levels ("/" -> ["", ""])."""
    return topic.split("/")


def matches(topic: str, topic_filter: str) -> bool:
    t, f = parse(topic), parse(topic_filter)
    i = 0
    while i < len(f):
        fl = f[i]
        if fl == "#":
            # last level only; zero or more remaining levels; never a '$' first level
            if i != len(f) - 1:
                return False
            if i == 0 and t[0].startswith("$"):
                return False
            return True  # i <= len(t) is guaranteed by the loop
        if i >= len(t):
            return False
        if fl == "+":
            if i == 0 and t[0].startswith("$"):
                return False
        elif fl != t[i]:
            return False
        i += 1
    return len(t) == len(f)
