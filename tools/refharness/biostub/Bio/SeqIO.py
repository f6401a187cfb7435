"""Stand-in for Bio.SeqIO.parse(handle_or_path, 'fasta'), restricted to what scripts/nr_flt.py touches: records with
`.id` (first word of the title line) and `.seq` (hashable, prints as the residues).  It follows Biopython's documented FASTA
rules (title = the '>' line without '>' and trailing blanks; sequence = the following lines, right-stripped and joined, with
blanks and carriage returns removed; text before the first '>' ignored).  The goldens made with it use only PLAIN FASTA
(one title word or word + description, residue lines without inner blanks, '\n' ends), on which every FASTA parser -- the
real Bio.SeqIO included -- yields the same (id, sequence) pairs: what those goldens pin is nr_flt.py's own logic."""


class _Record:
    def __init__(self, title, seq):
        words = title.split(None, 1)
        self.id = words[0] if words else ""
        self.description = title
        self.seq = seq


def parse(src, fmt):
    assert fmt == "fasta"
    handle = open(src) if isinstance(src, str) else src
    title, lines = None, []
    for line in handle:
        if line[:1] == ">":
            if title is not None:
                yield _Record(title, "".join(lines).replace(" ", "").replace("\r", ""))
            title, lines = line[1:].rstrip(), []
        elif title is not None:
            lines.append(line.rstrip())
    if title is not None:
        yield _Record(title, "".join(lines).replace(" ", "").replace("\r", ""))
